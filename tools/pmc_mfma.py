"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass: per kernel, the clock the chip held and the
share of cycles its matrix pipes were busy.

Usage: python tools/pmc_mfma.py <counter_collection.csv> [<out.txt>]

GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note): cycles = value / 8, clock = cycles /
dispatch wall time (reads high on dispatches under ~0.3 ms).  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs:
busy share = value / (1024 * cycles).  One fp32 32x32x2 MFMA holds its pipe for 64 cycles, so a kernel of F useful FLOPs
cannot show less than F / 4096 * 64 busy cycles."""
import collections
import csv
import sys


def main():
    src = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
    agg = collections.OrderedDict()
    seen = set()
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        e = agg.setdefault(k, {"n": 0, "t": 0.0, "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0})
        if r["Counter_Name"] in e:
            e[r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            e["n"] += 1
            e["t"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    lines = ["%-56s %5s %10s %9s %9s" % ("kernel", "n", "avg us", "clk GHz", "MFMA busy")]
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:24]:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or e["t"] <= 0:
            continue
        lines.append("%-56s %5d %10.1f %9.3f %8.1f%%" % (k[:56], e["n"], e["t"] / e["n"] / 1e3, cyc / e["t"],
                                                       100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)))
    for l in lines:
        print(l)
        if out:
            out.write(l + "\n")


if __name__ == "__main__":
    main()
