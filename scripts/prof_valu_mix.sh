# usage: bash scripts/prof_valu_mix.sh <outdir-name> <config>
# Per-kernel VALU instruction classes (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _INT64, _CVT) and the VALU-busy
# counters (SQ_ACTIVE_INST_VALU, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE) of `bench.py --config <config>`
# -> gpurun_out/<name>/valu_mix_<cfg>.txt / .json.  Counter passes only (no trace in the same run).
set -e
NAME=$1; CFG=$2
OUT=gpurun_out/$NAME
mkdir -p $OUT
c=$(echo $CFG | tr A-Z a-z)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python3 bench.py --config $CFG --no-cpu-baseline --no-early-out --no-tet --steps 3 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/mix1_$c -- $B > /dev/null 2> $OUT/mix1_$c.err || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/mix2_$c -- $B > /dev/null 2> $OUT/mix2_$c.err || true
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
mix = collections.defaultdict(dict)
for d in (f"mix1_{c}", f"mix2_{c}"):
    for k, v in load(d).items():
        mix[k].update(v)
json.dump(mix, open(f"{out}/valu_mix_{c}.json", "w"), indent=1, sort_keys=True)
with open(f"{out}/valu_mix_{c}.txt", "w") as f:
    for k, v in sorted(mix.items()):
        n = v.get("SQ_INSTS_VALU", 0.0)
        if n < 1e5:
            continue
        f.write(k + f": SQ_INSTS_VALU {n / 1e6:.2f} M\n")
        named = 0.0
        for cls in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT", "ADD_F64", "FMA_F64"):
            x = v.get("SQ_INSTS_VALU_" + cls, 0.0); named += x
            f.write(f"    {cls:10s} {x / 1e6:8.2f} M  {100 * x / n:5.1f} %\n")
        f.write(f"    {'other':10s} {(n - named) / 1e6:8.2f} M  {100 * (n - named) / n:5.1f} %   (moves, selects, compares, bit ops not in a class)\n")
        gui, act = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_ACTIVE_INST_VALU", 0.0)
        if gui:
            f.write(f"    GRBM_GUI_ACTIVE {gui:.0f} cycles; SQ_ACTIVE_INST_VALU {act / 1e6:.2f} M; VALUBusy = ACTIVE / 256 CUs / GUI = {act / 256 / gui:.3f}\n")
            f.write(f"    SQ_BUSY_CU_CYCLES {v.get('SQ_BUSY_CU_CYCLES', 0) / 1e6:.2f} M; SALU insts {v.get('SQ_INSTS_SALU', 0) / 1e6:.2f} M, SALU cycles {v.get('SQ_INST_CYCLES_SALU', 0) / 1e6:.2f} M, branches {v.get('SQ_INSTS_BRANCH', 0) / 1e6:.2f} M\n")
print(open(f"{out}/valu_mix_{c}.txt").read())
PY
