"""The gfx950 ISA of every kernel, linted at build time (no GPU): tools/asm_lint.py under pytest.

A round-1 wrong-result bug (accumulator read-back scheduled before the last MFMA of a K loop) was invisible to every
shape the numerics tests covered; the lint sees it in the instruction stream.  It also pins each matrix kernel to its
MFMA opcode (exact-fp32 32x32x2, or 32x32x16 bf16) and forbids scratch spills."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernels allowed to use scratch, with the reason.  Empty = none.
KNOWN_SCRATCH = {}       # round 2: the lint found 48-byte private arrays in three bf16 kernels (uint4 staging arrays SROA
#                          did not promote) -- fixed with native vector types, see conv_igemm_bf16.hip


@pytest.fixture(scope="module")
def rows():
    import asm_lint
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    return asm_lint.lint(jobs=min(8, os.cpu_count() or 4))


def test_every_kernel_compiles_and_matrix_kernels_are_found(rows):
    names = [r["kernel"] for r in rows]
    assert len(rows) > 100
    for stem in ("conv3x3_igemm_kernel<", "conv3x3_igemm_bf16_kernel<", "conv3x3_wgrad_kernel<", "pw_gemm_kernel<",
                 "conv3x3_igemm_lin_kernel<", "deconv_wgrad_kernel<", "conv3x3_wgrad_c3_kernel", "conv3x3_wgrad_bf16s_kernel",
                 "deconv_wgrad_bf16s_kernel", "deconv_wgrad_bf16s4_kernel", "deconv_wgrad4_kernel", "conv3x3_bf16s_kernel<"):
        assert any(stem in n for n in names), stem


def test_no_accumulator_read_back_inside_a_k_loop(rows):
    bad = [(r["kernel"], r["early_acc_reads"]) for r in rows if r["early_acc_reads"]]
    assert not bad, bad


def test_matrix_kernels_use_the_intended_mfma_opcode(rows):
    checked = [r for r in rows if r["expect"]]
    assert len(checked) >= 30
    bad = [(r["kernel"], r["expect"], r["opcodes"]) for r in checked if r["expect"] not in r["opcodes"]]
    assert not bad, bad
    # exact fp32 kernels must not mix in a reduced-precision pipe
    mixed = [r["kernel"] for r in checked if r["expect"].endswith("x2_f32") and any("bf16" in o or "f16" in o for o in r["opcodes"])]
    assert not mixed, mixed


def test_no_scratch_spills(rows):
    bad = [(r["kernel"], r["private_segment"], r["scratch_insts"]) for r in rows
           if (r["private_segment"] or r["scratch_insts"]) and not any(k in r["kernel"] for k in KNOWN_SCRATCH)]
    assert not bad, bad


def test_no_flat_loads(rows):
    """A load is `flat_` when the compiler cannot prove its address space -- round 5: a pointer select between a global tensor and a
    `__device__ const` zero page (constant address space).  Flat loads count on lgkmcnt as well, so every LDS read behind one waits
    for it: pw_gemm's A rows as flat loads made the headline step 27 % slower."""
    # pack_many_kernel reads its source / destination pointers from a table in device memory (unetk_pack_item): pointers loaded from
    # memory carry no address space, and neither an address_space(1) round trip nor __builtin_amdgcn_is_shared / is_private
    # assumptions made the compiler treat them as global (tried in round 5).  An HBM-bound 0.13 ms per step with no LDS traffic to
    # entangle with: left as it is.
    known = ("pack_many_kernel(",)
    bad = [(r["kernel"], r["flat_loads"]) for r in rows if r["flat_loads"] and not r["kernel"].startswith(known)]
    assert not bad, bad


# matrix kernels whose K loop prefetches through registers (global load -> VGPR -> LDS write behind the step's MFMAs)
PREFETCH_LOOPS = ("conv3x3_igemm_kernel<",)       # (pw_gemm keeps its branchy A-row loads: measured faster on the configuration in use, csrc/deconv.hip)
# the linear-pixel kernel's plain / stream-K / accumulate / half-chunk variants (its tap-subset and grouped-tap variants keep one early
# wait in their run-time tap loops: DESIGN.md 8)
LIN_PLAIN = "conv3x3_igemm_lin_kernel<"


def test_prefetch_waits_sit_behind_the_mfmas(rows):
    """profiles/r05_probe_lin_prefetch.txt: no s_waitcnt vmcnt right behind the request it waits for, in front of the MFMAs that
    were meant to hide it."""
    checked = [r for r in rows if any(r["kernel"].startswith(("void " + k, k)) for k in PREFETCH_LOOPS)]
    lin = [r for r in rows if LIN_PLAIN in r["kernel"] and ", false, false, " in r["kernel"].split(">")[0][len(LIN_PLAIN):]
           and r["kernel"].split("<")[1].split(",")[4].strip() == "false" and r["kernel"].split("<")[1].split(",")[8].strip() == "false"]
    assert len(checked) >= 20 and len(lin) >= 8, (len(checked), len(lin))
    bad = [(r["kernel"], r["early_vm_waits"]) for r in checked + lin if r["early_vm_waits"]]
    assert not bad, bad
