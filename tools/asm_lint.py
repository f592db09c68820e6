"""Lint the generated gfx950 ISA of every kernel in csrc/*.hip (cross-compiles with hipcc; no GPU needed).

Checks, per kernel:
1. **No accumulator read-back inside a K loop.**  Background (round 1): with two MFMA loops selected by a run-time flag
   in one kernel, the compiler kept the accumulator in VGPRs across tiles, copied it to AGPRs before each MFMA loop and
   read it back (`v_accvgpr_read`) right after -- with too few wait states after the last 16-pass MFMA, so rows 27 / 31
   of every 32-row panel lost the final k-step, non-reproducibly.  A healthy kernel reads its accumulators only AFTER
   the last MFMA in program text (the epilogue).
2. **The matrix kernels really are MFMA kernels of the intended shape**: fp32 kernels contain `v_mfma_f32_32x32x2_f32`
   (exact fp32), bf16 kernels `v_mfma_f32_32x32x16_bf16`; a template change that silently falls back to VALU FMAs (or to
   another MFMA shape with different rounding) is caught here, not by a slow benchmark.
3. **No scratch**: `.private_segment_fixed_size` == 0 and no `scratch_` instructions (a register spill in a hot loop).

Usage: python tools/asm_lint.py      (exits 1 on a finding).  `tests/test_asm_lint.py` runs the same checks under pytest.
"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "boxsegliver_amd", "csrc")

# kernel-name prefix -> MFMA opcode its body must contain (template instances checked individually)
EXPECT = [
    ("conv3x3_igemm_bf16_kernel", "v_mfma_f32_32x32x16_bf16"),
    ("pw_gemm_bf16_kernel", "v_mfma_f32_32x32x16_bf16"),
    ("conv3x3_igemm_kernel", "v_mfma_f32_32x32x2_f32"),
    ("conv3x3_igemm_lin_kernel", "v_mfma_f32_32x32x2_f32"),
    ("pw_gemm_kernel", "v_mfma_f32_32x32x2_f32"),
    ("conv3x3_wgrad_c3_kernel", "v_mfma_f32_32x32x2_f32"),
    ("conv3x3_c3_mfma_kernel", "v_mfma_f32_32x32x2_f32"),
    ("conv3x3_wgrad_c3_bf16s_kernel", "v_mfma_f32_32x32x2_f32"),     # fp32 image x bf16 dy: exact fp32 contraction
    ("conv3x3_wgrad_bf16s_kernel", "v_mfma_f32_32x32x16_bf16"),
    ("conv3x3_bf16s_kernel", "v_mfma_f32_16x16x32_bf16"),
    ("deconv_wgrad_bf16s_kernel", "v_mfma_f32_32x32x16_bf16"),
    ("deconv_wgrad_bf16s4_kernel", "v_mfma_f32_32x32x16_bf16"),
    ("deconv_wgrad4_kernel", "v_mfma_f32_32x32x2_f32"),
]


def expected_opcode(name):
    """The MFMA opcode a kernel of this (demangled) name must contain, or None when it is not a matrix kernel."""
    base = name.split("<")[0].split("(")[0].replace("void ", "").strip()
    for prefix, op in EXPECT:
        if base == prefix:
            return op
    if base in ("conv3x3_wgrad_kernel", "deconv_wgrad_kernel"):   # <.., BF16, ..> instances use the bf16 pipe
        args = name[name.find("<") + 1:name.rfind(">")].replace(" ", "").split(",") if "<" in name else []
        bf16 = (base == "conv3x3_wgrad_kernel" and len(args) > 2 and args[2] == "true") or \
               (base == "deconv_wgrad_kernel" and len(args) > 0 and args[0] == "true")
        return "v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x2_f32"
    return None


def _compile(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function", "-S",
                           "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def _early_vm_waits(body):
    """Barrier-to-barrier stretches (in layout order) with >= 8 MFMAs in which an `s_waitcnt vmcnt(..)` follows a global load of
    the SAME stretch before half of its MFMAs and within a quarter of them: the wave waits for a prefetch it has only just
    requested, in front of the matrix work that was meant to hide it.  Round 5 found this in every fp32 conv kernel: requests
    inside exec-masked blocks (`if (tid < ..) x = load`, `ok ? load : 0`), which the compiler closes with s_waitcnt vmcnt(0)."""
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    cuts = [-1] + [i for i, l in enumerate(lines) if l.startswith("s_barrier")] + [len(lines)]
    hits = 0
    for k in range(len(cuts) - 1):
        seg = lines[cuts[k] + 1:cuts[k + 1]]
        nm = sum(1 for l in seg if l.startswith("v_mfma"))
        if nm < 8:
            continue
        mf, loaded_at = 0, None
        for l in seg:
            if l.startswith("v_mfma"):
                mf += 1
            elif l.startswith(("global_load", "buffer_load")) and " lds" not in l:
                loaded_at = mf
            elif "vmcnt" in l and loaded_at is not None and mf < nm // 2 and mf - loaded_at < max(4, nm // 4):
                hits += 1
                break
    return hits


def lint(jobs=4):
    """[{file, kernel, n_mfma, early_acc_reads, opcodes, scratch_insts, private_segment, expect, findings}] over csrc."""
    rows = []
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        texts = list(pool.map(lambda s: _compile(s, tmp), srcs))
    for src, txt in zip(srcs, texts):
        private = {m.group(1): int(m.group(2)) for m in re.finditer(
            r"\.amdhsa_kernel (\w+)\b.*?\.amdhsa_private_segment_fixed_size (\d+)", txt, re.S)}
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n", txt, re.M):
            end = txt.find("s_endpgm", m.end())
            if end < 0:
                continue
            body = txt[m.end():end]
            sym = m.group(1)
            if sym not in private:
                continue                                     # a device function, not a kernel
            name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "")
            mf = [x.start() for x in re.finditer(r"v_mfma", body)]
            early = [x.start() for x in re.finditer(r"v_accvgpr_read", body) if mf and x.start() < mf[-1]]
            ops = sorted(set(re.findall(r"v_mfma_\w+", body)))
            scratch = len(re.findall(r"^\s*scratch_\w+", body, re.M))
            flat = len(re.findall(r"^\s*flat_load", body, re.M))
            early_waits = _early_vm_waits(body)
            want = expected_opcode(name)
            findings = []
            if early:
                findings.append("{} accumulator read(s) before the last MFMA".format(len(early)))
            if want and want not in ops:
                findings.append("expected {} but the body has {}".format(want, ops or "no MFMA"))
            if private[sym] > 0 or scratch:
                findings.append("scratch: private_segment {} B, {} scratch_ instruction(s)".format(private[sym], scratch))
            if flat:
                findings.append("{} flat_load(s): a load whose address space the compiler could not prove global".format(flat))
            rows.append(dict(file=os.path.basename(src), kernel=name, n_mfma=len(mf), early_acc_reads=len(early),
                             opcodes=ops, scratch_insts=scratch, private_segment=private[sym], expect=want,
                             flat_loads=flat, early_vm_waits=early_waits, findings=findings))
    return rows


def main():
    rows = lint()
    bad = 0
    for r in rows:
        if r["n_mfma"] or r["findings"]:
            print("%-84s mfma %4d  %s" % (r["kernel"][:84], r["n_mfma"], "; ".join(r["findings"]) or "ok"))
        bad += bool(r["findings"])
    print("{} kernels, {} with findings".format(len(rows), bad))
    print("FAIL" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
