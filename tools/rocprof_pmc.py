"""Diagnostic: sum of one PMC counter per kernel from a `rocprofv3 --pmc <counter> --kernel-trace` run (the sqlite database this ROCm writes).
usage: rocprof_pmc.py <dir-or-db>      prints kernel name, dispatches, counter name, sum over dispatches"""
import glob, os, sqlite3, sys
src = sys.argv[1]
dbs = [src] if os.path.isfile(src) else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    v = [n for n in names if n.startswith("counters_collection")]
    if not v:
        print("no counters_collection view in", db, names[:20]); continue
    cols = [r[1] for r in con.execute(f"pragma table_info({v[0]})")]
    kn = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    cv = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    for k, c, n, s in con.execute(f"select {kn}, {cn}, count(*), sum({cv}) from {v[0]} group by {kn}, {cn}"):
        print(f"{k.split('(')[0]},{n},{c},{s}")
