#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of the HIP engine (cross-compiles for gfx950; no GPU needed).

    python tools/kernel_resources.py [extra hipcc flags, e.g. -DMGPU_RECIP_MINWAVES=5] [--filter substr]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "maniac_mc_amd", "csrc", f) for f in ("mgpu_engine.hip", "mgpu_launch.hip", "mgpu_lanes.hip", "mgpu_windows.hip")]


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    err = ""
    with tempfile.TemporaryDirectory() as d:
        for src in SRCS:
            p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fopenmp", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                                "-o", os.path.join(d, "e.o")] + args, capture_output=True, text=True, cwd=d)
            if p.returncode != 0:
                sys.exit(p.stderr[-4000:])
            err += p.stderr

    class P:
        stderr = err
    p = P
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    names = [b.split("\n")[0].split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    keys = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("spillV", r"VGPR Spill"), ("spillS", r"SGPR Spill"),
            ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
    for b, name in zip(blocks, dem):
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace("mgpu::", "")
        if flt and flt not in name:
            continue
        vals = []
        for label, k in keys:
            m = re.search(k + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?':>4s}")
        print(f"{name:62s} " + "  ".join(vals))


if __name__ == "__main__":
    main()
