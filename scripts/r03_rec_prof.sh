mkdir -p gpurun_out/r3f
bash scripts/prof_valu_mix.sh r3f C4 > gpurun_out/r3f/mix.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f/trace -- python3 bench.py --config C4 --no-cpu-baseline --no-early-out --no-tet --steps 30 --warmup 5 > gpurun_out/r3f/bench_trace.json 2> gpurun_out/r3f/trace.err
cp gpurun_out/r3f/trace/*/*_kernel_stats.csv gpurun_out/r3f/kernel_stats_c4.csv
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r3f/pmcx -- python3 bench.py --config C4 --no-cpu-baseline --no-early-out --no-tet --steps 3 --warmup 2 > /dev/null 2> gpurun_out/r3f/pmcx.err
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r3f/pmcx/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        if "tri_" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {n: round(sum(x)/len(x)/1e6, 2) for n, x in sorted(v.items())}, "launches", len(next(iter(v.values()))))
PY
cut -d, -f1-4 gpurun_out/r3f/kernel_stats_c4.csv | head -12
tail -3 gpurun_out/r3f/bench_trace.json | cut -c1-600
