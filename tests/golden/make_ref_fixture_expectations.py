"""Expected results for the reference's own reader fixtures (tests/golden/ref_fixtures/, copied DATA
files of /root/reference/tests/readers/**), produced by the reference's own front end.

Runs only where oracle/_ref exists (needs /root/reference + amdflang at build time).  For every
fixture it runs oracle/dump_ref_files.py in a subprocess (the reference allocates once / `stop`s on
bad input) and records either the dumped state (.npz) or the stop code.

    python tests/golden/make_ref_fixture_expectations.py
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIX = os.path.join(HERE, "ref_fixtures")


def write_parameters_inc():
    """parameters.inc is not in the snapshot (it lives in the un-fetched mc-topology submodule):
    rebuild it from the `Pair Coeffs` section of good-01.data, one `pair_coeff i i eps sigma` per type."""
    lines = open(os.path.join(FIX, "good-01.data")).read().splitlines()
    i = lines.index("Pair Coeffs # lj/cut/coul/long") + 2
    out = []
    while lines[i].strip():
        t, e, s = lines[i].split()[:3]
        out.append(f"pair_coeff {t} {t} {e} {s}")
        i += 1
    open(os.path.join(FIX, "parameters.inc"), "w").write("\n".join(out) + "\n")


def run(maniac, data, inc, stage, out_npz):
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "dump_ref_files.py"), maniac, data, inc,
                            tmp + "/out/", str(stage), out_npz], capture_output=True, text=True, cwd=tmp)
    ok = "DUMP_OK" in p.stdout
    msg = ""
    for ln in (p.stdout + p.stderr).splitlines():
        if "ERROR STOP" in ln or "FATAL" in ln or "STOP: code" in ln or "Error" in ln:
            msg = msg + ln.strip() + " | "
    return ok, p.returncode, msg[:300]


def main():
    write_parameters_inc()
    inc = os.path.join(FIX, "parameters.inc")
    summary = {}
    for f in sorted(os.listdir(FIX)):
        if f.endswith(".data"):
            ok, rc, msg = run(os.path.join(FIX, "input.maniac"), os.path.join(FIX, f), inc, 2,
                              os.path.join(FIX, f.replace(".data", ".expected.npz")))
            summary[f] = dict(ok=ok, returncode=rc, message=msg)
    for f in sorted(os.listdir(os.path.join(FIX, "inputs"))):
        if f.endswith(".maniac"):
            ok, rc, msg = run(os.path.join(FIX, "inputs", f), "unused", "unused", 1,
                              os.path.join(FIX, "inputs", f.replace(".maniac", ".expected.npz")))
            summary["inputs/" + f] = dict(ok=ok, returncode=rc, message=msg)
    json.dump(summary, open(os.path.join(FIX, "expected_outcomes.json"), "w"), indent=1)
    for k, v in summary.items():
        print(k, v["ok"], v["returncode"], v["message"][:80])


if __name__ == "__main__":
    main()
