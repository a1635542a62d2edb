"""Debug helper: the timeline of one step of the default schedule from a rocprofv3 kernel trace
(tools/collect_profiles.sh keeps it under gpurun_out/prof_rNN/default/): start, end, duration (ms from the step's
bin_kernel), hardware queue, stream, workgroups, LDS per workgroup and name of every kernel of at least 0.1 ms.
usage: step_timeline.py <kernel_trace.csv> [step index, default: the last but one]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    wg = int(r["Workgroup_Size_X"]) or 1
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"], r["Stream_Id"], int(r["Grid_Size_X"]) // wg, r["LDS_Block_Size"]))
ks.sort()
starts = [s for s, e, n, *_ in ks if n.startswith("bin_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
t0 = starts[k]
t1 = starts[k + 1] if k + 1 < len(starts) else 1 << 62
print(f"step {k} of {len(starts)}: {(min(t1, max(e for s, e, *_ in ks)) - t0) / 1e6:.1f} ms")
for s, e, n, q, st, g, lds in ks:
    if t0 <= s < t1 and e - s > 1e5:
        print(f"{(s - t0) / 1e6:8.1f} {(e - t0) / 1e6:8.1f} {(e - s) / 1e6:7.1f}  q{q} s{st} wg{g:>6} lds{lds:>7}  {n}")
