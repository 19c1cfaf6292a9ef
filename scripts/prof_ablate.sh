# VALU instruction counts of k_tri_backward / k_tri_forward under ablation flags
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# the DMR_ABLATE switches exist in the ablation build only (python -m dmesh_renderer_amd.build --ablation)
export DMR_LIBRARY=$GRAFT_REPO_ROOT/dmesh_renderer_amd/libdmesh_renderer_hip_ablation.so
for a in "$@"; do
  rm -rf gpurun_out/abl_$a
  DMR_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/abl_$a/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tet > /dev/null 2>&1
  echo "ABLATE=$a"; python3 scripts/pmc_summary.py gpurun_out/abl_$a | grep "tri_backward\|tri_forward"
done
