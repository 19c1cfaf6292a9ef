# usage: bash scripts/prof_shard.sh <outdir-name> [config] [partition] ; rocprofv3 kernel traces of ONE rank's work when the frame (default C4)
# is sharded 1 / 2 / 4 / 8 ways (bench.py --emulate-rank R/N: the rank's tile-row band, one process, no collective): first, middle and
# last rank of every split -> kernel sums per rank, and the gradient all-reduce's payload
set -e
OUT=gpurun_out/$1
CFG=${2:-C4}
PART=${3:-auto}
SPECS=${SPECS:-"0/1 0/2 1/2 0/4 2/4 3/4 0/8 3/8 7/8"}
STEPS=30; if [ "$CFG" = "C5" ]; then STEPS=8; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for spec in $SPECS; do
  tag=$(echo $spec | tr / _)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 bench.py --config $CFG --emulate-rank $spec --partition $PART --steps $STEPS --warmup 3 > $OUT/bench_$tag.json 2> $OUT/trace_$tag.err || true
  cp $OUT/trace_$tag/*/*_kernel_stats.csv $OUT/kernel_stats_rank_$tag.csv
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(f"{out}/kernel_stats_rank_*.csv")):
    tag = f.split("rank_")[1][:-4]
    # (kernels of the steady state only: k_scan_hits runs once, in the first backward, which has no size estimate yet)
    rows = [r for r in csv.DictReader(open(f)) if "dmr::" in r["Name"] and int(r["Calls"]) >= 6]
    per = {r["Name"].split("(")[0].replace("void ", "").replace("dmr::", "").split("<")[0]: round(float(r["AverageNs"]) / 1000, 1) for r in rows}
    b = json.load(open(f"{out}/bench_{tag}.json"))
    c = b["config"]
    fixed = sum(v for k, v in per.items() if k in ("k_project_verts", "k_setup_faces_lds", "k_scan_tiles", "k_scatter_faces_lds", "k_tri_unpack",
                                                   "k_scan_tiles_partial", "k_scan_tiles_blocks", "k_scan_tiles_final", "k_scan_hits"))
    res[tag] = {"kernel_sum_us": round(sum(per.values()), 1), "fixed_part_us": round(fixed, 1), "ms_per_step": b["ms_per_step"], "band": c["parallelism"],
                "allreduce_payload_bytes": c.get("allreduce_payload_bytes"), "kernels_us": per}
    print(tag, res[tag]["kernel_sum_us"], "us kernels;", b["ms_per_step"], "ms/step;", per)
json.dump(res, open(f"{out}/shard_kernel_sums.json", "w"), indent=1)
PY
