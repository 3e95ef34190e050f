#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, workgroups): calls, avg / min / max duration in microseconds.
   python tools/trace_by_grid.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y") or 1) * int(r.get("Workgroup_Size_Z") or 1))
    wgs = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1) // wg
    agg[(name, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, w), v in sorted(agg.items()):
    print("%-44s wgs=%6d calls=%4d avg=%9.2f min=%9.2f max=%9.2f" % (n[:44], w, len(v), sum(v) / len(v), min(v), max(v)))
