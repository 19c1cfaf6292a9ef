# usage: bash scripts/final_profiles.sh <name> ; everything profiles/<round>/ holds for one build: traces / counters / traffic of C4 and C3,
# per-rank kernel sums of the sharded C4 step, the bench lines of C1-C4 (+ --sync at C4) and the module timings
set -e
N=$1
mkdir -p gpurun_out/$N
bash scripts/prof_all.sh $N C4 > gpurun_out/$N/prof_c4.log 2>&1
bash scripts/prof_all.sh $N C3 > gpurun_out/$N/prof_c3.log 2>&1
bash scripts/prof_shard.sh $N > gpurun_out/$N/prof_shard.log 2>&1
for c in C1 C2 C3 C4; do python bench.py --config $c > gpurun_out/$N/bench_$(echo $c | tr A-Z a-z).json 2> /dev/null; done
python bench.py --sync --no-cpu-baseline > gpurun_out/$N/bench_c4_sync.json 2> /dev/null
for c in C1 C2 C3 C4; do python scripts/time_module.py $c gpurun_out/$N/time_module_$c.json > /dev/null 2>&1; done
tail -22 gpurun_out/$N/prof_c4.log; tail -8 gpurun_out/$N/prof_shard.log; cut -c1-330 gpurun_out/$N/bench_c4.json
