# Round-end measurement, part D: SQ counters of one hm355_ctu_kernel launch over 320 4K pictures (two --pmc passes of 8 counters, --kernel-trace only).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r03_sq1 -o s -- python3 $R/tools/quick_timing.py 3840 2160 320 > $R/gpurun_out/r03_sq1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/r03_sq2 -o s -- python3 $R/tools/quick_timing.py 3840 2160 320 > $R/gpurun_out/r03_sq2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, json
tot = {}
for d in ("$R/gpurun_out/r03_sq1", "$R/gpurun_out/r03_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0] == "hm355_ctu_kernel":
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
wc = tot.get("SQ_WAVE_CYCLES", 1.0); ctus = 2040 * 320
out = {"workload": "tools/quick_timing.py 3840 2160 320: one launch of hm355_ctu_kernel over 320 4K 10-bit pictures (652,800 CTUs; 256 workgroups of twelve searches), two rocprofv3 --pmc passes (8 SQ counters each) with --kernel-trace",
       "counters": tot,
       "fraction_of_SQ_WAVE_CYCLES": {k: tot[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in tot},
       "wave_instructions_per_ctu": {k: tot[k] / ctus for k in tot if k.startswith("SQ_INSTS")}}
json.dump(out, open("$R/gpurun_out/r03_pmc_sq_summary.json", "w"), indent=1)
print(json.dumps(out["fraction_of_SQ_WAVE_CYCLES"]), json.dumps(out["wave_instructions_per_ctu"]))
PY
