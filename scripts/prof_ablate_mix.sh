# usage: bash scripts/prof_ablate_mix.sh <outdir-name> <bits>...   (ablation build; VALU instruction classes per DMR_ABLATE setting)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/libdmesh_renderer_hip_ablation.so
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for a in "$@"; do
  DMR_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/abl_$a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tet --no-early-out > /dev/null 2> $OUT/abl_$a.err
  python3 - "$OUT/abl_$a" "$a" <<'PY' | tee -a $OUT/ablate_mix.txt
import collections, csv, glob, sys
d, a = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "k_tri_forward" in k or "k_tri_backward" in k:
            agg[k][r["Counter_Name"].replace("SQ_INSTS_", "")].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"ABLATE={a:>6s} {k:28s} " + "  ".join(f"{n} {sum(x) / len(x) / 1e6:7.2f}" for n, x in sorted(v.items())))
PY
done
