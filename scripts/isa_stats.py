"""Static instruction counts and register use of one kernel in a `hipcc -S --cuda-device-only` listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o tri.s csrc/dmr_tri.hip
    python scripts/isa_stats.py tri.s k_tri_backward_hits
"""
import re
import sys


def stats(path, kernel):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in lines[start + 1:end]]
    ins = [l for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
    out = {"total": len(ins)}
    for pre in ("v_", "s_", "ds_", "global_", "scratch_", "buffer_"):
        out[pre.rstrip("_")] = sum(l.startswith(pre) for l in ins)
    txt = "\n".join(lines[end:end + 60])
    for key in ("num_vgpr", "numbered_sgpr", "private_seg_size"):
        m = re.search(r"\.%s, (\d+)" % key, txt)
        if m:
            out[key] = int(m.group(1))
    return out


if __name__ == "__main__":
    print(sys.argv[2], stats(sys.argv[1], sys.argv[2]))
