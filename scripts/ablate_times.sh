# usage: bash scripts/ablate_times.sh <name> <bits>... ; bench stage times of the ABLATION build under DMR_ABLATE=<bits>
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/libdmesh_renderer_hip_ablation.so
for a in "$@"; do
  echo "ABLATE=$a" >> $OUT/ablate.txt
  DMR_ABLATE=$a python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-early-out --no-tet 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])" >> $OUT/ablate.txt || exit 1
done
cat $OUT/ablate.txt
