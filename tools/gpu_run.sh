#!/usr/bin/env bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mix
rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -w tools/mfma_mix.hip -o /tmp/mfma_mix || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -o mix -- /tmp/mfma_mix > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_mix")
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-12:], r["Grid_Size"])
    e = d.setdefault(key, {})
    e[r["Counter_Name"]] = float(r["Counter_Value"]); e["t"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_mix_summary.txt"), "w")
for k, e in d.items():
    cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
    if e["t"] < 2e6: continue
    line = "%s grid %s  t %.2f ms  clk %.3f GHz  mfma_busy %.1f %%" % (k[1], k[2], e["t"] / 1e6, cyc / e["t"], 100 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc))
    print(line); out.write(line + "\n")
PY
rm -rf $OUT
