# usage: bash scripts/prof_all.sh <outdir-name> <config> [extra bench.py flags...]
# rocprofv3 kernel trace + two SQ counter passes + the two HBM traffic passes of `bench.py --config <config>`, summarised into
#   kernel_stats_<cfg>.csv  pmc_summary_<cfg>.txt  pmc_<cfg>.json  traffic_<cfg>.json  bench_<cfg>.json
# under gpurun_out/<name>/ (copy what backs DESIGN.md to profiles/<round>/).  Counters and traces are separate runs.
set -e
NAME=$1; CFG=$2; shift; shift
OUT=gpurun_out/$NAME
mkdir -p $OUT
c=$(echo $CFG | tr A-Z a-z)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python3 bench.py --config $CFG --no-cpu-baseline --no-early-out --no-tet $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -- $B --steps 30 --warmup 5 > $OUT/bench_trace_$c.json 2> $OUT/trace_$c.err || true
cp $OUT/trace_$c/*/*_kernel_stats.csv $OUT/kernel_stats_$c.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc1_$c.err || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc2_$c.err || true
# lane utilisation: active VALU threads per issued VALU instruction = gfx950's derived VALUUtilization
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc3_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc3_$c.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcf_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/fetch_$c.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcw_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/write_$c.err || true
python3 - "$OUT" "$c" <<'PY'
import collections, csv, glob, json, sys
out, c = sys.argv[1], sys.argv[2]
def load(d):
    fs = glob.glob(f"{out}/{d}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("dmr::"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in agg.items()}
pmc = collections.defaultdict(dict)
for d in (f"pmc1_{c}", f"pmc2_{c}", f"pmc3_{c}"):
    for k, v in load(d).items():
        if d.startswith("pmc3"):  # the third pass: only what the first two do not hold
            v = {n: x for n, x in v.items() if n == "SQ_THREAD_CYCLES_VALU"} | {"SQ_ACTIVE_INST_VALU_pass3": v.get("SQ_ACTIVE_INST_VALU", 0.0)}
        pmc[k].update(v)
for k, v in pmc.items():  # VALUUtilization (counter_defs.yaml): thread-cycles / (instruction-cycles x 64), both from the same pass
    if v.get("SQ_THREAD_CYCLES_VALU") and v.get("SQ_ACTIVE_INST_VALU_pass3"):
        v["valu_lane_util"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU_pass3"] * 64.0)
json.dump(pmc, open(f"{out}/pmc_{c}.json", "w"), indent=1, sort_keys=True)
with open(f"{out}/pmc_summary_{c}.txt", "w") as f:
    for k, v in sorted(pmc.items()):
        f.write(k.ljust(30) + " " + str({n: round(x / 1e6, 2) for n, x in sorted(v.items())}) + " (x1e6)\n")
        if v.get("SQ_WAVE_CYCLES"):
            f.write(" " * 31 + f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES']:.3f}\n")
        if v.get("valu_lane_util"):
            f.write(" " * 31 + f"VALU lane utilisation (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU) = {v['valu_lane_util']:.3f}\n")
tr = collections.defaultdict(dict)
for name, d in (("fetch", f"pmcf_{c}"), ("write", f"pmcw_{c}")):
    for k, v in load(d).items():
        tr[k][name + "_raw_kb"] = list(v.values())[0]
for k, v in tr.items():
    f_, w_ = v.get("fetch_raw_kb", 0.0), v.get("write_raw_kb", 0.0)
    # gfx950: FETCH_SIZE counts 128-B fabric reads as 64 B -> doubled (calibrated for wide streaming reads only; these kernels
    # gather, so the doubled figure is an upper estimate); WRITE_SIZE is exact.  Units: KiB.
    v["hbm_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
    v["hbm_bytes_per_launch_uncorrected"] = (f_ + w_) * 1024.0
json.dump(tr, open(f"{out}/traffic_{c}.json", "w"), indent=1, sort_keys=True)
for r in list(csv.DictReader(open(f"{out}/kernel_stats_{c}.csv")))[:12]:
    print(r["Name"].split("(")[0][-34:].ljust(36), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1000:9.1f} us')
for k, v in sorted(tr.items()):
    print(k.ljust(34), round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB (uncorrected", round(v["hbm_bytes_per_launch_uncorrected"] / 1e6, 1), ")")
PY
grep "WAIT_ANY /" -B1 $OUT/pmc_summary_$c.txt | grep -v "^--" | cut -c1-60 | paste - - | grep "tri_\|tet_" || true
