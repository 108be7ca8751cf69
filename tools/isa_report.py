#!/usr/bin/env python3
"""Disassembly of the pair sweep's hot loop (cross-compiles for gfx950; no GPU needed).

    python tools/isa_report.py [--kernel SUBSTR] [-D...] > profiles/rNN/pair_sweep_isa.txt

Compiles maniac_mc_amd/csrc/mgpu_engine.hip to gfx950 assembly (`hipcc -S --cuda-device-only`), cuts out the kernel whose
mangled name contains SUBSTR (default: pair_sweep_kernel<3, false, false, true, true>, the fused SPC/E sweep bench.py
times), finds its hot basic block -- the ALL_C unit body: the block with the most fp64 VALU instructions -- prints it and
counts its instructions by class, per unit (64 atoms x NREG site-states) and per site-atom term, beside the PMC figure of
the whole kernel (SQ_INSTS_VALU per wave-term, profiles/rNN/pmc_kernels_spce.txt).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "maniac_mc_amd", "csrc", "mgpu_engine.hip")

def main():
    args = sys.argv[1:]
    want = "pair_sweep_kernelILi3ELb0ELb0ELb1ELb1E"
    if "--kernel" in args:
        k = args.index("--kernel")
        want = args[k + 1]
        del args[k:k + 2]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "e.s")
        p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fopenmp", "-S", "--cuda-device-only", "-o", out, SRC] + args,
                           capture_output=True, text=True)
        if p.returncode != 0:
            sys.exit(p.stderr[-3000:])
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if want in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel"))
    name = subprocess.run(["c++filt", lines[start].split(":")[0]], capture_output=True, text=True).stdout.strip()
    meta = {}
    for l in lines[end:end + 60]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
        if m:
            meta[m.group(1)] = m.group(2)
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        else:
            t = l.split(";")[0].strip()
            if t and not t.startswith("."):
                cur[1].append(t)
    blocks.append(cur)

    def n_f64(ins):
        return sum(1 for i in ins if re.match(r"v_\w+_f64", i))
    hot = max(blocks, key=lambda b: n_f64(b[1]))
    ins = hot[1]
    # the straight-line unit body ends at the first branch of the block (the rare r < 0.5 A path follows it)
    cut = next((k for k, i in enumerate(ins) if i.startswith("s_and_saveexec") or i.startswith("s_cbranch")), len(ins))
    body = ins[:cut]
    n_lds = sum(1 for i in body if i.startswith("ds_read"))
    terms = n_lds // 3                                   # three 16-byte reads per site-atom term
    print(f"# {name}")
    print(f"# VGPRs {meta.get('next_free_vgpr')}, SGPRs {meta.get('next_free_sgpr')}, scratch {meta.get('private_segment_fixed_size')} B/lane, "
          f"static LDS {meta.get('group_segment_fixed_size')} B; {len(blocks)} basic blocks, {sum(len(b[1]) for b in blocks)} instructions")
    print(f"# hot block {hot[0]}: {len(body)} instructions before its first branch, {terms} site-atom terms per unit of 64 atoms\n")
    counts = {}

    def bump(k):
        counts[k] = counts.get(k, 0) + 1
    masked = set()                      # VGPRs holding the high word of a row start (s with its low mantissa bits cleared)
    for i in body:
        op = i.split()[0]
        ops = i[len(op):]
        base = re.sub(r"_e(32|64)$", "", op)
        if base == "v_and_b32":
            m = re.match(r"\s*v(\d+),", ops)
            if m:
                masked.add(int(m.group(1)))
        if base in ("v_add_f64",):
            m = re.search(r"-v\[(\d+):(\d+)\]\s*$", ops)
            if "|" in ops:
                bump("fold: L - |d|            v_add_f64 s, -|v|")
            elif m and int(m.group(2)) in masked:
                bump("table: t = s - s_row     v_add_f64")
            else:
                bump("separation: xj - rx ...  v_add_f64")
        elif base == "v_min_f64":
            bump("fold: min(|d|, L - |d|)  v_min_f64")
        elif base in ("v_mul_f64",):
            bump("r^2: first square        v_mul_f64")
        elif base in ("v_fmac_f64", "v_fma_f64"):
            bump("v_fmac_f64 / v_fma_f64 (2 per term finish r^2, 6 per term the degree-6 polynomial, 1 per term the accumulate when in this block)")
        elif base == "v_cvt_f64_f32":
            bump("polynomial: fp32 coefficients c5, c6 to fp64   v_cvt_f64_f32")
        elif base in ("v_ashrrev_i32", "v_subrev_u32", "v_min_u32", "v_mad_u32_u24", "v_bfe_u32", "v_lshl_add_u32", "v_sub_u32", "v_lshrrev_b32"):
            bump(f"table index              {base}")
        elif base == "v_and_b32":
            bump("table: high word of s_row (mantissa bits cleared)  v_and_b32")
        elif base in ("v_or_b32", "v_or3_b32") or base.startswith("v_cmp"):
            bump(f"below-table test (one per 2-3 terms)  {base}")
        elif base.startswith("ds_read"):
            bump("LDS: table row           ds_read_b128")
        elif base.startswith("s_waitcnt"):
            bump("s_waitcnt")
        elif base.startswith("v_"):
            bump(f"other VALU               {base}")
        else:
            bump(f"other                    {base}")
    valu = sum(v for k, v in counts.items() if not k.startswith(("LDS", "s_waitcnt", "other   ")))
    print("| class | per unit | per term |\n|---|---|---|")
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {v} | {v / max(1, terms):.2f} |")
    print(f"| **VALU total** | {valu} | **{valu / max(1, terms):.2f}** |\n")
    # the rest of one loop iteration: the blocks from the hot one to the back edge
    hi = blocks.index(hot)
    print("## the loop around it (blocks up to the back edge; VALU / SALU / memory instructions, and how each ends)\n")
    print("| block | instructions | VALU | SALU | memory | ends with |\n|---|---|---|---|---|---|")
    for b in blocks[hi:hi + 48]:
        br = [i for i in b[1] if "branch" in i]
        print(f"| {b[0]} | {len(b[1])} | {sum(1 for i in b[1] if i.startswith('v_'))} | {sum(1 for i in b[1] if i.startswith('s_'))} | "
              f"{sum(1 for i in b[1] if i.startswith(('global_', 'scratch_', 'ds_', 'buffer_')))} | {'; '.join(br[-2:]) if br else 'falls through'} |")
        if any(hot[0] in i for i in br):
            break
    print()
    print("## the block\n")
    for i in body:
        print("    " + i)


if __name__ == "__main__":
    main()
