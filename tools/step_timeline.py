"""Print the kernel timeline of the last complete train step from a rocprofv3 --kernel-trace CSV
(start offset from the step's first kernel, duration, queue, workgroups, kernel name).

    python tools/step_timeline.py gpurun_out/<dir>/<prefix>_kernel_trace.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the encoder pool (SAIL) / the token gather (ARK): print the last complete one
anchor = next((a for a in ("pool_gather_fwd", "enc_pool_fwd", "tok_gather") if any(a in r["Kernel_Name"] for r in rows)), "tok_gather")
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r['Queue_Id']} wg{wg:<6d} {r['Kernel_Name'][:70]}")
