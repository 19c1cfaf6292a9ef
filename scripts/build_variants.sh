# usage: bash scripts/build_variants.sh name1:"-DFLAG=.." name2:"..." ; tuning builds of the C ABI library under dmesh_renderer_amd/variants/
# (run one with DMR_LIBRARY=dmesh_renderer_amd/variants/lib_<name>.so)
cd "$(dirname "$0")/.."
mkdir -p dmesh_renderer_amd/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-function -Wl,-rpath,/opt/rocm/lib $flags \
    -o dmesh_renderer_amd/variants/lib_$name.so dmesh_renderer_amd/csrc/dmr_api.hip dmesh_renderer_amd/csrc/dmr_binning.hip dmesh_renderer_amd/csrc/dmr_tri.hip dmesh_renderer_amd/csrc/dmr_tet.hip &
done
wait
ls -la dmesh_renderer_amd/variants/
