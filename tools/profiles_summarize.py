"""Turn rocprofv3 CSV output into the small text summaries kept under profiles/ (see tools/collect_profiles.sh)."""
import collections
import csv
import json
import sys

bench_trace, single_trace, fetch_csv, write_csv, out = sys.argv[1:6]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def by_grid(path, dst, header):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        wgs = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
        agg[(short(r["Kernel_Name"]), wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(dst, "w") as f:
        f.write(header)
        for (k, w) in sorted(agg, key=lambda kw: (kw[0], -kw[1])):
            v = agg[(k, w)]
            f.write("%-28s wgs=%6d calls=%4d avg=%9.2f min=%9.2f max=%9.2f\n" % (k, w, len(v), sum(v) / len(v), min(v), max(v)))


import os as _os
if _os.path.exists(bench_trace):
  by_grid(bench_trace, out + "/bench_kernel_by_grid.txt",
        "# rocprofv3 --kernel-trace, per (kernel, workgroups): calls avg_us min_us max_us\n"
        "# command: python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (contexts in flight overlap: durations include co-running kernels)\n")
by_grid(single_trace, out + "/single_kernel_by_grid.txt",
        "# rocprofv3 --kernel-trace, per (kernel, workgroups): calls avg_us min_us max_us\n"
        "# command: python tools/single_stream.py 20   (one context, uncontended kernel durations)\n")


def counters(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        agg[(short(r["Kernel_Name"]), wgs)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


F, Wr = counters(fetch_csv, "FETCH_SIZE"), counters(write_csv, "WRITE_SIZE")
rows = []
for k in F:
    f_kb, w_kb = F[k], Wr.get(k, 0.0)
    rows.append((k[0], k[1], f_kb, w_kb, 2.0 * f_kb * 1024 + w_kb * 1024))
rows.sort(key=lambda r: -r[4])
with open(out + "/pmc_hbm_traffic.txt", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python tools/single_stream.py 3\n"
            "# units: KB as reported; read bytes = 2 x FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section); per launch averages\n"
            "%-28s %6s %14s %14s %16s\n" % ("kernel", "wgs", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "HBM_bytes(2F+W)"))
    for r in rows:
        f.write("%-28s %6d %14.0f %14.0f %16.0f\n" % r)
blur0 = [r[4] for r in rows if r[0].startswith("k_blur") and ", false," in r[0] and r[1] >= 900]
import hashlib
import os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha1()
for _f in ("pyramid.hip", "blur_arith.h"):
    _h.update(open(os.path.join(_root, "popsift_amd", "csrc", "hip", _f), "rb").read())
json.dump({"k_blur_octave0_hbm_bytes_per_launch": sum(blur0) / max(1, len(blur0)), "n_launch_kinds": len(blur0),
           "kernel_sources_sha1": _h.hexdigest(),        # bench.py reports the traffic only while the kernel's sources still hash to this
           "source": "pmc_hbm_traffic.txt of the same tools/collect_profiles.sh run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2xFETCH correction)"},
          open(out + "/pmc_summary.json", "w"), indent=1)
