"""Lint the generated gfx950 ISA of the MFMA kernels for accumulator read-back inside a K loop.

Background (round 1): with two MFMA loops selected by a run-time flag in one kernel, the compiler kept the accumulator
in VGPRs across tiles, copied it to AGPRs before each MFMA loop and read it back (`v_accvgpr_read`) right after -- with
too few wait states after the last 16-pass MFMA, so rows 27 / 31 of every 32-row panel lost the final k-step,
non-reproducibly.  A healthy kernel reads its accumulators only AFTER the last MFMA in program text (the epilogue).

Usage: python tools/asm_lint.py      (cross-compiles csrc/*.hip to ISA with hipcc; no GPU needed; exits 1 on a hit)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "boxsegliver_amd", "csrc")


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function", "-S",
                                   "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
            txt = open(out).read()
            for m in re.finditer(r"^(_Z\w+):[^\n]*\n", txt, re.M):
                body = txt[m.end():txt.find("s_endpgm", m.end())]
                mf = [x.start() for x in re.finditer(r"v_mfma", body)]
                if not mf:
                    continue
                early = [x.start() for x in re.finditer(r"v_accvgpr_read", body) if x.start() < mf[-1]]
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "")
                print("%-84s mfma %4d  accumulator reads before the last mfma: %d" % (name[:84], len(mf), len(early)))
                bad += bool(early)
    print("FAIL" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
