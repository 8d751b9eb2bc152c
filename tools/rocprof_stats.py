"""Diagnostic: per-kernel summary (calls, total / average / min / max duration, LDS, scratch, grid) of a `rocprofv3 --kernel-trace` run whose
output is the sqlite database this ROCm writes.   usage: rocprof_stats.py <dir-or-db> <out.csv>"""
import csv, glob, os, sqlite3, statistics, sys
src, out = sys.argv[1], sys.argv[2]
dbs = [src] if os.path.isfile(src) else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
rows = {}
for db in dbs:
    con = sqlite3.connect(db)
    views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table') and name like 'kernels%'")]
    for v in views[:1]:
        for name, start, end, lds, scratch, gx, wx in con.execute(f"select name, start, end, lds_size, scratch_size, grid_x, workgroup_x from {v}"):
            rows.setdefault(name.split("(")[0], []).append((end - start, lds, scratch, gx, wx))
vg = {}
if not dbs:                            # --output-format csv: <prefix>_kernel_trace.csv carries the same columns plus the register counts
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0]
            rows.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
            vg[n] = (int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"]))
tot = sum(sum(d[0] for d in v) for v in rows.values()) or 1
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "LDS", "Scratch", "Grid", "Workgroup", "VGPR", "AccumVGPR", "SGPR"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
        d = [x[0] for x in v]
        w.writerow([name, len(d), sum(d), sum(d) / len(d), 100.0 * sum(d) / tot, min(d), max(d), statistics.pstdev(d), v[0][1], v[0][2], v[0][3], v[0][4]] + list(vg.get(name, ("", "", ""))))
print(open(out).read())
