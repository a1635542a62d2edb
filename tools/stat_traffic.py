"""HBM traffic of one statistics pass from the rocprofv3 PMC files of tools/collect_profiles.sh:
    python tools/stat_traffic.py r03  ->  profiles/r03_stat_traffic.json
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md), summed over the kernels of the
statistics set and divided by the number of passes; the algorithmic bytes come from the bench line of the same round."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAT_KERNELS = ("bin_kernel", "stat_plan_kernel", "stat_lanes", "stat_lean_kernel", "set_kernel<0")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name).replace("void ", "")


def per_pass(path, counter):
    tot, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] != counter or not k.startswith(STAT_KERNELS):
            continue
        tot[k] += float(r["Counter_Value"])
        calls[k] += 1
    passes = calls[next(k for k in calls if k.startswith("bin_kernel"))]
    return {k: v / passes for k, v in tot.items()}, passes


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r03"
    prof = os.path.join(ROOT, "profiles")
    fetch, passes = per_pass(os.path.join(prof, f"{r}_stat_pmc_fetch_size.csv"), "FETCH_SIZE")
    write, _ = per_pass(os.path.join(prof, f"{r}_stat_pmc_write_size.csv"), "WRITE_SIZE")
    bench = json.load(open(os.path.join(prof, f"{r}_bench_default.json")))
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    fk, wk = sum(fetch.values()), sum(write.values())
    hbm = int(2 * fk * 1024 + wk * 1024)
    out = {"source": f"{r}_stat_pmc_fetch_size.csv, {r}_stat_pmc_write_size.csv ({passes} passes)",
           "rule": "HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950), summed over the statistics set's kernels",
           "FETCH_SIZE_KiB_per_pass": fk, "WRITE_SIZE_KiB_per_pass": wk,
           "per_kernel_KiB": {k: {"fetch": fetch.get(k, 0.0), "write": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write))},
           "hbm_bytes_per_pass": hbm, "algorithmic_bytes_per_pass": alg, "ratio": hbm / alg}
    json.dump(out, open(os.path.join(prof, f"{r}_stat_traffic.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_pass", "algorithmic_bytes_per_pass", "ratio")}))


if __name__ == "__main__":
    main()
