#!/usr/bin/env python3
"""Per-stage latency of one step / one window of the single-chain drop-in, from a rocprofv3 trace made by
tools/trace_chain.sh (t_hip_api_trace.csv, t_kernel_trace.csv of `tools/chain_speed.py --cases ... --ks K`).

    python tools/chain_latency.py <trace dir> [<trace dir> ...] > profiles/rNN/chain_latency.md

The steady-state part of the run (the middle half of the engine-call sequence) is cut into periods at every host
"wait" -- hipStreamSynchronize in the batched path, or the gap the host spends polling the pinned tag in the one-launch
path (no API call: the period boundary is the chain kernel's launch) -- and the median period is reported: what the host
calls, how long each call takes, which kernels run and for how long, and what is left (host loop, queueing, polling).
Under the tracer every HIP call costs ~1-2 us more than untraced; the untraced step rate is printed by chain_speed.py.
"""
import csv
import os
import statistics
import sys


def load(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def short(name):
    name = name.replace("void mgpu::", "").replace("mgpu::", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def analyse(d):
    api = [r for r in load(os.path.join(d, "t_hip_api_trace.csv")) if r["Domain"].startswith("HIP_RUNTIME")]
    ker = load(os.path.join(d, "t_kernel_trace.csv"))
    for r in api + ker:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    api.sort(key=lambda r: r["s"])
    ker.sort(key=lambda r: r["s"])
    chain = any("chain_window_kernel" in r["Kernel_Name"] for r in ker)
    if chain:
        marks = [r["s"] for r in ker if "chain_window_kernel" in r["Kernel_Name"]]
        marks_api = sorted(r["s"] for r in api if r["Function"] in ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel"))
    else:
        marks = [r["e"] for r in api if r["Function"] == "hipStreamSynchronize"]
    lo, hi = len(marks) // 4, 3 * len(marks) // 4
    marks = marks[lo:hi]
    periods = []
    for a, b in zip(marks[:-1], marks[1:]):
        calls = [r for r in api if a <= r["s"] < b and r["Function"] not in ("hipSetDevice", "hipGetLastError")]
        kers = [r for r in ker if a <= r["s"] < b]
        periods.append((b - a, calls, kers))
    if not periods:
        return None
    # the most common shape (same sequence of kernels), median over its periods
    shapes = {}
    for p in periods:
        key = tuple(short(k["Kernel_Name"]) for k in p[2])
        shapes.setdefault(key, []).append(p)
    out = []
    total = len(periods)
    for key, ps in sorted(shapes.items(), key=lambda kv: -len(kv[1]))[:3]:
        med = statistics.median(p[0] for p in ps) / 1e3
        rows = []
        n_calls = min(len(p[1]) for p in ps)
        for i in range(n_calls):
            names = {p[1][i]["Function"] for p in ps}
            if len(names) != 1:
                break
            rows.append(("host call", names.pop(), statistics.median((p[1][i]["e"] - p[1][i]["s"]) for p in ps) / 1e3))
        for i, kname in enumerate(key):
            rows.append(("kernel", kname, statistics.median((p[2][i]["e"] - p[2][i]["s"]) for p in ps) / 1e3))
        gpu_busy = statistics.median(sum(k["e"] - k["s"] for k in p[2]) for p in ps) / 1e3
        api_busy = statistics.median(sum(c["e"] - c["s"] for c in p[1]) for p in ps) / 1e3
        out.append(dict(share=len(ps) / total, period_us=med, rows=rows, gpu_busy_us=gpu_busy, api_busy_us=api_busy))
    return dict(dir=d, chain=chain, shapes=out, n_periods=total)


def main():
    print("# Single-chain latency: where one step / one window goes\n")
    print(__doc__.split("\n\n")[2].strip() + "\n")
    for d in sys.argv[1:]:
        res = analyse(d)
        print(f"## `{d}`\n")
        if res is None:
            print("no steady-state periods found\n")
            continue
        print(f"path: {'one launch per window (mgpu_chain_window)' if res['chain'] else 'batched submit / wait / commit'}; "
              f"{res['n_periods']} periods in the steady-state half of the run\n")
        for sh in res["shapes"]:
            print(f"### period shape seen in {100 * sh['share']:.0f} % of the periods: median {sh['period_us']:.1f} us "
                  f"(GPU busy {sh['gpu_busy_us']:.1f} us, host inside HIP calls {sh['api_busy_us']:.1f} us)\n")
            print("| stage | what | median us |\n|---|---|---|")
            for kind, name, us in sh["rows"]:
                print(f"| {kind} | `{name}` | {us:.1f} |")
            print()


if __name__ == "__main__":
    main()
