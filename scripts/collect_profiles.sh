#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline` block (run on the GPU box from the repo root):
#   scripts/collect_profiles.sh <tag>      -> gpurun_out/<tag>/{kernel_stats.csv,bench_train_only_under_rocprof.json,traffic.json,mfma.json}
# Four separate passes (PMC counters never share a pass with --stats; FETCH_SIZE and WRITE_SIZE need a pass each).
set -o pipefail
R=$(pwd); T=${1:-prof}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-only > $O/bench_train_only_under_rocprof.json 2> $O/stats.err || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --train-only > $O/f.json 2> $O/f.err || exit 1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --train-only > $O/w.json 2> $O/w.err || exit 1
echo "write done"
python3 $R/scripts/collect_traffic.py $O/f $O/w $O/traffic.json > $O/traffic.txt || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/m -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --train-only > $O/m.json 2> $O/m.err || exit 1
python3 $R/scripts/collect_mfma.py $O/m $O/mfma.json > $O/mfma.txt || exit 1
echo "mfma done"
# the raw counter dumps are large: keep the summaries only
rm -rf $O/stats $O/f $O/w $O/m
head -12 $O/mfma.txt; head -8 $O/traffic.txt
