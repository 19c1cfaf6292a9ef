"""Copy what scripts/final_profiles.sh left under gpurun_out/<name>/ into profiles/<round>/ (the files DESIGN.md and bench.py cite).

    python scripts/copy_profiles.py f2 r02
"""
import glob, json, os, shutil, sys

out, dst = os.path.join("gpurun_out", sys.argv[1]), os.path.join("profiles", sys.argv[2])
os.makedirs(dst, exist_ok=True)
sums = json.load(open(f"{out}/shard_kernel_sums.json"))
json.dump(sums, open(f"{dst}/shard_kernel_sums_c4.json", "w"), indent=1)
for f in sorted(glob.glob(f"{out}/kernel_stats_rank_*.csv")):
    shutil.copy(f, f"{dst}/kernel_stats_c4_rank_{f.split('rank_')[1]}")
for c in ("c4", "c3"):
    for n in (f"kernel_stats_{c}.csv", f"pmc_{c}.json", f"pmc_summary_{c}.txt", f"traffic_{c}.json"):
        shutil.copy(f"{out}/{n}", f"{dst}/{n}")
for n in ["bench_c1.json", "bench_c2.json", "bench_c3.json", "bench_c4.json", "bench_c4_sync.json"] + [f"time_module_C{i}.json" for i in (1, 2, 3, 4)]:
    shutil.copy(f"{out}/{n}", f"{dst}/{n}")
print({k: v["kernel_sum_us"] for k, v in sums.items()})
