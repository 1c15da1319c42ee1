#!/bin/bash
# The round's profile set, on the GPU box (one rocprofv3 pass per counter group: gpurun refuses --pmc combined with the
# trace domains other than --kernel-trace, and FETCH_SIZE / WRITE_SIZE do not fit one pass). Results under gpurun_out/prof/.
# usage: tools/profile_round.sh [tag]      (the program itself follows `--`: no env / bash -c hop under the profiler)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
TAG=${1:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
find_csv() { find "$1" -name "*$2" | head -1; }
echo "== warm the box (a fresh box pages the image in during its first minute: not part of any figure)"; python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config4 --no-in-flight > /dev/null 2>&1
echo "== bench line"; python3 $REPO/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.log || exit 1
echo "== kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config4 --no-in-flight > $OUT/${TAG}_stats_run.json 2> $OUT/stats.log || exit 1
cp "$(find_csv $OUT/stats kernel_stats.csv)" $OUT/${TAG}_kernel_stats_bench.csv
python3 $REPO/tools/prof_summary.py $OUT/${TAG}_kernel_stats_bench.csv 27 52 > $OUT/${TAG}_kernel_stats_summary_per_proof.txt
echo "== SQ counters"; rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --hbm-resident --no-config4 --no-in-flight > /dev/null 2> $OUT/sq.log || exit 1
python3 $REPO/tools/pmc_summary.py "$(find_csv $OUT/sq counter_collection.csv)" > $OUT/${TAG}_pmc_sq.txt
python3 $REPO/tools/valu_from_pmc.py "$(find_csv $OUT/sq counter_collection.csv)" $OUT/${TAG}_valu.json > /dev/null
echo "== clock + VALU rate"; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/clk -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --hbm-resident --no-config4 --no-in-flight > /dev/null 2> $OUT/clk.log || exit 1
python3 $REPO/tools/clock_from_pmc.py "$(find_csv $OUT/clk counter_collection.csv)" > $OUT/${TAG}_clock_valu.txt
echo "== FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --hbm-resident --no-config4 --no-in-flight > /dev/null 2> $OUT/fetch.log || exit 1
echo "== WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --hbm-resident --no-config4 --no-in-flight > /dev/null 2> $OUT/write.log || exit 1
python3 $REPO/tools/traffic_from_pmc.py "$(find_csv $OUT/fetch counter_collection.csv)" "$(find_csv $OUT/write counter_collection.csv)" $OUT/${TAG}_traffic.json > /dev/null
echo "== timeline"; rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $REPO/tools/trace_run.py 20 notrace > /dev/null 2> $OUT/tl.log || exit 1
python3 $REPO/tools/timeline.py "$(find_csv $OUT/tl kernel_trace.csv)" > $OUT/${TAG}_timeline_gaps.txt
echo "== config 4"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bb -- python3 $REPO/bench.py --config babybear --steps 5 --warmup 2 --no-cpu-baseline --no-config4 --no-in-flight > $OUT/${TAG}_config4_stats_run.json 2> $OUT/bb.log || exit 1
cp "$(find_csv $OUT/bb kernel_stats.csv)" $OUT/${TAG}_config4_kernel_stats.csv
python3 $REPO/tools/prof_summary.py $OUT/${TAG}_config4_kernel_stats.csv 14 30 > $OUT/${TAG}_config4_kernel_stats_summary_per_proof.txt
python3 $REPO/bench.py --config babybear --steps 20 --warmup 5 > $OUT/${TAG}_config4_bench_line.json 2> $OUT/bb_line.log || exit 1
echo "== host-resident timeline"; rocprofv3 --kernel-trace --output-format csv -d $OUT/tlh -- python3 $REPO/tools/trace_host.py > /dev/null 2> $OUT/tlh.log || exit 1
python3 $REPO/tools/timeline_host.py "$(dirname "$(find_csv $OUT/tlh kernel_trace.csv)")" 60 > $OUT/${TAG}_timeline_host.txt
echo "== joint vs plain"; python3 $REPO/tools/joint_vs_plain.py > $OUT/${TAG}_joint_vs_plain.txt 2>&1 || exit 1
python3 $REPO/tools/joint_kernel_diff.py > $OUT/${TAG}_joint_kernel_diff.txt 2>&1 || exit 1
echo "== micro benchmarks (the source of the integer issue peak and of the field-arithmetic rates)"
for m in b3_rate gl_sgpr gl_rate acc_rate pull_rate; do
  if [ -x $REPO/tools/micro/$m ]; then timeout -k 10 120 $REPO/tools/micro/$m > $OUT/${TAG}_micro_$m.txt 2>&1 || echo "micro $m failed" >> $OUT/${TAG}_micro_$m.txt; fi
done
echo "== host narrowing: consecutive rows against runs of rows (the row-group experiment of DESIGN section 0)"
g++ -O2 -std=c++17 -pthread $REPO/tools/micro/pack_runs.cpp $REPO/multi-stark_amd/csrc/pack_host.cpp -o /tmp/pack_runs 2> $OUT/pack_runs_build.log &&
  { echo "# tools/micro/pack_runs.cpp on this box's host: 16 threads, then 32"; timeout -k 10 120 /tmp/pack_runs 16; echo; timeout -k 10 120 /tmp/pack_runs 32; } > $OUT/${TAG}_micro_pack_runs.txt 2>&1
rm -rf $OUT/stats $OUT/sq $OUT/clk $OUT/fetch $OUT/write $OUT/tl $OUT/bb $OUT/tlh
ls -la $OUT
