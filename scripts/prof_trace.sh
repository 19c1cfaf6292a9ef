# usage: bash scripts/prof_trace.sh <name> ; writes gpurun_out/<name>/{kernel_stats.csv,bench.json}
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err || true
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-160
