"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in
KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced (16 B / lane) streaming
read, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Separate
passes (FETCH_SIZE takes 3 of the 4 TCC slots)."""
import collections
import csv
import json
import sys


def agg(path, cname):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d = out.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    fa, wa = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    res = {}
    for k, (n, fs) in fa.items():
        wn, ws = wa.get(k, [n, 0.0])
        f_b, w_b = fs / n * 1024.0, ws / max(wn, 1) * 1024.0
        res[k] = {"launches": n, "fetch_size_bytes_per_launch_raw": f_b, "write_size_bytes_per_launch": w_b,
                  "hbm_bytes_per_launch_corrected": 2.0 * f_b + w_b}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of "
                       "`bench.py --steps 2 --warmup 1`; read side x2 per the gfx950 FETCH_SIZE correction",
               "kernels": res}, open(dst, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches"])[:12]:
        print("%-48s n %4d  HBM/launch %8.1f MB (read x2 %8.1f, write %8.1f)" % (
            k[:48], v["launches"], v["hbm_bytes_per_launch_corrected"] / 1e6,
            2 * v["fetch_size_bytes_per_launch_raw"] / 1e6, v["write_size_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
