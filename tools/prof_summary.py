"""Summarise a rocprofv3 rocpd database: per-kernel call count / total / avg / min / max (us)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
bygrid = len(sys.argv) > 2 and sys.argv[2] == "grid"
q = ("select name, %s count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
     "from kernels group by name %s order by 3 desc") % (("grid_x/workgroup_x,", ", grid_x") if bygrid else ("0,", ""))
rows = cur.execute(q).fetchall()
tot = sum(r[3] for r in rows)
print("%-64s %7s %6s %12s %10s %10s %10s %6s" % ("kernel", "wgs", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-64s %7d %6d %12.1f %10.2f %10.2f %10.2f %6.1f" % (name[:64], r[1], r[2], r[3], r[4], r[5], r[6], 100 * r[3] / tot))
