# usage: bash scripts/prof_trace_tet.sh <name> ; rocprofv3 kernel trace of the tet renderer at C3 -> gpurun_out/<name>/kernel_stats_tet.csv
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tet -- python3 scripts/time_tet.py --steps 20 > $OUT/tet_c3.json 2> $OUT/trace_tet.err || true
cp $OUT/trace_tet/*/*_kernel_stats.csv $OUT/kernel_stats_tet.csv
head -8 $OUT/kernel_stats_tet.csv | cut -c1-120
