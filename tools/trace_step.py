#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV of a training run and print the LAST step dispatch by dispatch:
kernel, workgroups, duration, gap to the previous dispatch.  Usage: trace_step.py kernel_trace.csv [n_last_steps]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at k_ssm_prep
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ssm_prep") or "k_ssm_prep" in r["Kernel_Name"]]
if len(starts) < 2:
    print("no step boundaries"); sys.exit(1)
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); tend = int(rows[b]["Start_Timestamp"])
print(f"step: {len(step)} dispatches, wall {(tend - t0)/1e3:.1f} us, sum of kernel time {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)/1e3:.1f} us")
prev_end = t0
agg = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * (int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * int(r["Grid_Size_Z"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    gap = (s - prev_end) / 1e3
    if "-v" in sys.argv:
        print(f"{(s - t0)/1e3:9.1f} {name[:48]:48s} wg {wg:6d} lds {int(r['LDS_Block_Size']):6d} vgpr {int(r['VGPR_Count']):3d}+{int(r['Accum_VGPR_Count']):3d} {(e - s)/1e3:8.1f} us gap {gap:6.1f}")
    k = agg.setdefault(name, [0, 0.0, 0.0])
    k[0] += 1; k[1] += (e - s) / 1e3; k[2] += max(gap, 0.0)
    prev_end = max(prev_end, e)
print(f"{'kernel':52s} {'n':>4s} {'us':>9s} {'gap us':>8s}")
for name, (n, us, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:52]:52s} {n:4d} {us:9.1f} {gap:8.1f}")
print("total gap us", sum(v[2] for v in agg.values()))
