"""Diagnostic: profiles/rNN_traffic.json from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the guide prescribes) of
`python3 bench.py --steps S --warmup W --no-cpu-baseline`.  The record carries the build id of the library that ran, so bench.py quotes it only for that build.
usage: make_traffic_json.py <fetch_dir> <write_dir> <out.json> <frames> <lanes> <kernel> [width height]"""
import csv, glob, json, os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hm-16.2_amd")]
import hm355


def counter(src, kernel, name):
    """(sum of the counter over the kernel's dispatches in KB, dispatches)"""
    tot, ids = 0.0, set()
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0] == kernel and r["Counter_Name"] == name:
                tot += float(r["Counter_Value"]); ids.add((f, r["Dispatch_Id"]))
    return tot, len(ids)


a = sys.argv
fetch_dir, write_dir, out, frames, lanes, kernel = a[1], a[2], a[3], int(a[4]), int(a[5]), a[6]
w, h = (int(a[7]), int(a[8])) if len(a) > 8 else (3840, 2160)
f_kb, nf = counter(fetch_dir, kernel, "FETCH_SIZE")
w_kb, nw = counter(write_dir, kernel, "WRITE_SIZE")
assert nf and nw and nf == nw, (nf, nw)
ctus = ((w + 63) // 64) * ((h + 63) // 64) * frames
rd, wr = f_kb * 1024 / nf, w_kb * 1024 / nw
lib = hm355.load_library()
rec = {"width": w, "height": h, "frames": frames, "lanes": lanes, "build_id": lib.hm355_build_id().decode(), "kernel": kernel, "launches": nf,
       "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_bytes_per_launch": rd + wr,
       "per_ctu_bytes": {"read": rd / ctus, "written": wr / ctus},
       "note": "FETCH_SIZE / WRITE_SIZE are KB of L2 memory-side requests (Infinity-Cache hits included); raw sums over the kernel's launches divided by "
               "the launches, as in rounds 1 and 2 (the guide's x2 correction of FETCH_SIZE applies to 16 B/lane streaming reads; this kernel's "
               "accesses are 2-8 B/lane block rows and register save areas)."}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
