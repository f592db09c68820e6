"""Summarise a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE pass: per kernel, the share of LDS-active cycles
lost to bank conflicts and the LDS utilisation.  Usage: python tools/pmc_lds.py <counter_collection.csv> [<out.txt>]
(GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles = value / 8; the SQ counters are summed over the chip: per CU = / 256.)"""
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
        e = agg.setdefault(k, collections.defaultdict(float))
        e[r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            e["n"] += 1
            e["t"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    lines = ["%-58s %5s %9s %12s %14s" % ("kernel", "n", "avg us", "LDS util", "conflict share")]
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:20]:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or e["SQ_LDS_IDX_ACTIVE"] <= 0:
            continue
        lines.append("%-58s %5d %9.1f %11.1f%% %13.1f%%" % (k[:58], e["n"], e["t"] / e["n"] / 1e3,
                                                         100.0 * e["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc),
                                                         100.0 * e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]))
    for l in lines:
        print(l)
        if out:
            out.write(l + "\n")


if __name__ == "__main__":
    main()
