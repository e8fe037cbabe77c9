#!/bin/bash
# Everything profiles/rNN_* is made from, on ONE GPU box (gpurun -- 'bash scripts/collect_profiles.sh TAG'):
# bench lines of every workload, rocprofv3 kernel stats of the C3 bench, PMC traffic (per conv op and per whole
# step) and matrix-pipe counters.  Outputs under gpurun_out/<TAG>_*; copy the ones to keep into profiles/.
set -o pipefail
TAG=${1:-r03}
O=gpurun_out
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
( time python3 bench.py > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench.err ) 2> $O/${TAG}_bench_c3_walltime.txt || exit 1; echo "c3 bench done"
for w in c2 c4 c5 r50v2 c5fp32; do python3 bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2>> $O/${TAG}_bench.err || exit 1; done
python3 bench.py --mode eval --workload c2 --no-cpu-baseline > $O/${TAG}_bench_c2_eval.json 2>> $O/${TAG}_bench.err || exit 1
python3 bench.py --mode eval --no-cpu-baseline > $O/${TAG}_bench_c3_eval.json 2>> $O/${TAG}_bench.err || exit 1
MVG_SPLIT=0 python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_c3_fp32mfma.json 2>> $O/${TAG}_bench.err || exit 1
# the captured step (single-stream hipGraph replay): host time per step vs the eager step
for w in c3 c4 c2; do python3 bench.py --workload $w --graph --no-cpu-baseline --no-roofline > $O/${TAG}_bench_${w}_graph.json 2>> $O/${TAG}_bench.err || exit 1; done
python3 scripts/graph_probe.py c4 2>&1 | grep -v amdgpu.ids > $O/${TAG}_graph_probe_c4.txt
echo "benches done"
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-overlap > /dev/null 2>> $O/${TAG}_bench.err || exit 1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" $O/${TAG}_c3_rocprofv3_kernel_stats.csv
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_conv_$c && CONV_PASS_LOG=/tmp/conv_pass_ops.json rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_conv_$c -- python3 scripts/conv_pass.py 50 128 4 split > /dev/null 2>> $O/${TAG}_bench.err || exit 1
  rm -rf /tmp/pmc_step_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_step_$c -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-overlap > /dev/null 2>> $O/${TAG}_bench.err || exit 1
done
python3 scripts/pmc_traffic.py /tmp/pmc_conv_FETCH_SIZE /tmp/pmc_conv_WRITE_SIZE c3 $O/${TAG}_pmc_traffic_c3.json split > /dev/null || exit 1
python3 scripts/pmc_summarize.py traffic /tmp/pmc_step_FETCH_SIZE /tmp/pmc_step_WRITE_SIZE 2 $O/${TAG}_pmc_step_traffic_c3.json || exit 1
python3 scripts/pmc_per_op.py /tmp/pmc_conv_FETCH_SIZE /tmp/pmc_conv_WRITE_SIZE /tmp/conv_pass_ops.json $O/${TAG}_pmc_per_conv_op_c3.txt || echo "per-op join failed"
# ... and over whole C5 (bf16 path) steps
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_step5_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_step5_$c -- python3 bench.py --workload c5 --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-overlap > /dev/null 2>> $O/${TAG}_bench.err || exit 1
done
python3 scripts/pmc_summarize.py traffic /tmp/pmc_step5_FETCH_SIZE /tmp/pmc_step5_WRITE_SIZE 2 $O/${TAG}_pmc_step_traffic_c5.json || exit 1
rm -rf /tmp/prof_stats5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-overlap > /dev/null 2>> $O/${TAG}_bench.err || exit 1
cp "$(find /tmp/prof_stats5 -name '*kernel_stats.csv' | head -1)" $O/${TAG}_c5_rocprofv3_kernel_stats.csv
echo "traffic done"
rm -rf /tmp/pmc_mfma && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pmc_mfma -- python3 scripts/conv_pass.py 50 128 4 split > /dev/null 2>> $O/${TAG}_bench.err || exit 1
python3 scripts/pmc_summarize.py mfma /tmp/pmc_mfma $O/${TAG}_pmc_mfma_util_c3.json || exit 1
python3 scripts/conv_bench.py 50 128 4 10 split 2>&1 | grep -v amdgpu.ids > $O/${TAG}_conv_shapes_r50_c3_split.txt
python3 scripts/bn_bench.py 128 4 split 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bn_shapes_c3_split.txt
echo "all done"
