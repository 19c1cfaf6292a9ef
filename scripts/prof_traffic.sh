# HBM traffic per launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), per MI355X_MICROARCH.md
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-early-out --no-tet > /dev/null 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-early-out --no-tet > /dev/null 2> $OUT/write.err || true
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for name in ("fetch", "write"):
    fs = glob.glob(f"{out}/pmc_{name}/*/*_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("dmr::"):
            agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][name + "_raw_kb"] = sum(v) / len(v)
for k, v in res.items():
    f, w = v.get("fetch_raw_kb", 0.0), v.get("write_raw_kb", 0.0)
    # gfx950: FETCH_SIZE counts 128-B fabric reads as 64 B -> doubled (calibrated for wide streaming reads only;
    # these kernels gather, so the doubled figure is an upper estimate); WRITE_SIZE is exact.  Units: KiB.
    v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    v["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024.0
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(k.ljust(34), {a: round(b / 1e6, 2) for a, b in v.items() if a.startswith("hbm")}, "MB")
PY
