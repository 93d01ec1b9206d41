#!/bin/bash
# Profiles of a round on the GPU box (see profiles/README.md).  One script for every round since round 5 (it replaces
# profile_round.sh / profile_round3.sh / profile_round4.sh of rounds 1-4):
#
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r05 [step ...]'
#
# Steps (default: all, in this order): ranks8 bench stats forward timeline pmc fps knn2d hotpath rehearsal corrclock knngate
# (ranks8 first: eight ranks of the default bench on this box's ONE GPU, on the fresh box's cold MIOpen database -- start-up account)
# Output: gpurun_out/<round>/prof/ ; copy what is to be judged into profiles/<round>_*.
# PMC passes are separate runs, counters only with --kernel-trace, the program itself behind "--" (no env / bash -c hops).
set -x
R=${1:-r05}; shift
STEPS=${*:-ranks8 bench stats forward timeline pmc fps knn2d hotpath rehearsal corrclock knngate}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/prof
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
S="python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py"

pmc() {  # tag passes... -- command...   (passes: a b c d e, see below)
  tag=$1; shift
  passes=()
  while [ "$1" != "--" ]; do passes+=($1); shift; done
  shift
  for pass in "${passes[@]}"; do
    case $pass in
      a) C="FETCH_SIZE";;
      b) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
      c) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
      d) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS";;
      e) C="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES";;
      f) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE";;
    esac
    rm -rf /tmp/pmc_${tag}_$pass
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$pass -- "$@" > /tmp/pmc_${tag}_$pass.log 2>&1)
  done
}

if has ranks8; then  # functional 8-rank run (all ranks on the one GPU, collective on gloo): time to the first step per rank, cold database
  ( time timeout 900 python3 bench.py --gpus 8 --share-gpu --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-corr-microbench --eval-batches 8 > $OUT/bench_8rank_shared_gpu.json 2> $OUT/bench_8rank_shared_gpu.err ) 2> $OUT/bench_8rank_shared_gpu_time.txt
fi
if has bench; then  # the bench lines themselves, un-profiled (first process on this box)
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  python3 bench.py --workload eval --steps 128 --no-cpu-baseline --no-corr-microbench > $OUT/bench_eval.json 2> $OUT/bench_eval.err
  python3 bench.py --config dsec > $OUT/bench_dsec.json 2> $OUT/bench_dsec.err
  python3 bench.py --workload eval --steps 128 --no-cpu-baseline --no-corr-microbench --eval-raw-events 300000 > $OUT/bench_eval_raw_events.json 2> $OUT/bench_eval_raw_events.err
fi
if has stats; then  # the default bench command, kernel stats
  (cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --eval-batches 0 --backend none > $OUT/bench_under_rocprof.json 2> /tmp/p1.err)
  cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
fi
if has forward; then  # one eager forward: launches / step and per-kernel totals
  (cd /tmp && rm -rf /tmp/p2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_forward.py 3 > /tmp/p2.log 2>&1)
  python3 tools/trace_window.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) 3 > $OUT/forward_kernel_stats.txt
fi
if has timeline; then  # stream timeline of the replayed graph
  python3 tools/stamp_timeline.py > $OUT/forward_stream_timeline.txt 2>/dev/null
fi
if has pmc; then
  pmc knn16 a b c -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py knn16 6
  pmc pointconv a b c -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py pointconv 6
  pmc corr a b c d e -- python3 $GRAFT_REPO_ROOT/tools/prof_corr.py 0 6
  pmc pw a b c -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py pw_l1 8
  $S knn_mfma_kernel $OUT/knn16_pmc.json "k_nearest_neighbor 3-D k=16, B=8, 8192 -> 4096: the sweep kernel (tools/prof_ops.py knn16 6)" /tmp/pmc_knn16_a /tmp/pmc_knn16_b /tmp/pmc_knn16_c
  $S knn_tie_replay_kernel $OUT/knn16_replay_pmc.json "the same search: its tied rows' second launch (tools/prof_ops.py knn16 6)" /tmp/pmc_knn16_a /tmp/pmc_knn16_b /tmp/pmc_knn16_c
  $S pointconv_fused_kernel $OUT/pointconv_pmc.json "PointConvNoSampling 195->128, B=4, N=4096 (tools/prof_ops.py pointconv 6)" /tmp/pmc_pointconv_a /tmp/pmc_pointconv_b /tmp/pmc_pointconv_c
  $S pointwise_conv_kernel $OUT/pw_l1_pmc.json "1x1 convolution 96 -> 510 over 4 x 144 x 240, the level-1 cross block's project_in (tools/prof_ops.py pw_l1 8)" /tmp/pmc_pw_a /tmp/pmc_pw_b /tmp/pmc_pw_c
  python3 tools/experiments/write_bw.py > $OUT/pw_l1_memory_floors.txt 2>/dev/null
  $S corr_mfma_dma_kernel $OUT/corr_microbench_pmc.json "correlation2d 1x256x544x960 md=4 fp32 (tools/prof_corr.py 0 6)" /tmp/pmc_corr_a /tmp/pmc_corr_b /tmp/pmc_corr_c /tmp/pmc_corr_d /tmp/pmc_corr_e
fi
if has fps; then
  pmc fps a b f -- python3 $GRAFT_REPO_ROOT/tools/prof_fps.py 5
  $S fps_pruned2_kernel $OUT/fps_pmc.json "furthest_point_sampling 8 x 8192 -> 4096 (tools/prof_fps.py 5)" /tmp/pmc_fps_a /tmp/pmc_fps_b /tmp/pmc_fps_f
fi
if has knn2d; then  # the nearest-point searches of the five levels: binned against the sweeps, kernel times
  pmc knn2d a b c -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py knn2d 6
  $S nearest2d_search_kernel $OUT/knn2d_search_pmc.json "k_nearest_neighbor 2-D k=1, B=8, 4096 points, 144x240 raster queries: the search kernel (tools/prof_ops.py knn2d 6)" /tmp/pmc_knn2d_a /tmp/pmc_knn2d_b /tmp/pmc_knn2d_c
  (cd /tmp && rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 $GRAFT_REPO_ROOT/tools/knn2d_bench.py > $OUT/knn2d_bench.txt 2>/dev/null)
  cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $OUT/knn2d_kernel_stats.csv
fi
if has hotpath; then  # the hot-path sequence alone, eager, kernel stats; and which generic ATen kernels are left in it / in the forward
  (cd /tmp && rm -rf /tmp/p6 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -- python3 $GRAFT_REPO_ROOT/bench.py --workload hotpath --eager --steps 10 --warmup 2 --no-cpu-baseline --no-corr-microbench --backend none > /tmp/p6.log 2>&1)
  cp $(find /tmp/p6 -name "*kernel_stats.csv" | head -1) $OUT/hotpath_process_kernel_stats.csv   # (the whole process: model construction and warm-up included)
  (cd /tmp && rm -rf /tmp/p7 && rocprofv3 --kernel-trace --output-format csv -d /tmp/p7 -- python3 $GRAFT_REPO_ROOT/tools/prof_forward.py 10 hotpath > /tmp/p7.log 2>&1)
  TRACE_TOP=200 python3 tools/trace_window.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 10 > $OUT/hotpath_kernel_stats.txt   # per STEP: between two marker kernels
  python3 bench.py --workload hotpath --no-cpu-baseline --no-corr-microbench > $OUT/bench_hotpath.json 2> $OUT/bench_hotpath.err
  python3 tools/aten_gpu_census.py hotpath > $OUT/aten_census_hotpath.txt 2>/dev/null
  python3 tools/aten_gpu_census.py forward > $OUT/aten_census_forward.txt 2>/dev/null
fi
if has rehearsal; then  # the 8-rank host side: host-only ranks; a real rank alone and among seven paced host-side neighbours, cached set pinned (the default of bench.py's eval leg) and pageable (staged by loader threads)
  timeout -k 5 300 python3 tools/host_rehearsal.py --ranks 8 --batches 384 2>/dev/null | grep "^{" > $OUT/host_rehearsal_default.json
  for mode in pinned pageable; do
    flag=""; [ $mode = pinned ] && flag="--pinned"
    timeout -k 5 600 python3 tools/host_rehearsal.py --ranks 1 --batches 256 --real-rank 0 $flag 2>/dev/null | grep "^{" > $OUT/host_rehearsal_real_alone_$mode.json
    timeout -k 5 600 python3 tools/host_rehearsal.py --ranks 8 --batches 256 --real-rank 0 --neighbours cpu --pace 66 $flag 2>/dev/null | grep "^{" > $OUT/host_rehearsal_real_cpu_neighbours_$mode.json
  done
fi
if has corrclock; then  # engine clock over the very launches of the correlation microbench, per operand kind
  python3 tools/corr_clock.py --out $OUT/corr_clock.json > $OUT/corr_clock.log 2>&1
fi
if has knngate; then
  python3 tools/knn_gate_table.py > $OUT/knn_gate_table.txt 2>/dev/null
fi
ls -la $OUT
