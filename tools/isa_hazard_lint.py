#!/usr/bin/env python3
"""ISA hazard lint for the hand-written gfx950 kernels.

The kernels hold ~20 multi-instruction inline-asm blocks (DPP reductions, the depthwise stencil, packed complex
arithmetic, `relu` reading MFMA accumulators ...).  The compiler's hazard recognizer does not look inside inline asm and
the hardware does not interlock these cases, so a missing wait state is silent: round 2's branch-free block-4 epilogue read
stale accumulators until `s_nop 15; s_nop 3` was spelled out.  This lint compiles a translation unit to ISA
(`hipcc --offload-device-only -S`) and walks every function's instruction stream, checking the data hazards the code
relies on:

  R1  a VGPR written by v_mfma_* must not be read by a VALU / LDS / memory instruction within the wait states its pass
      count demands (the number is CALIBRATED per opcode: a two-instruction kernel is compiled and the s_nops the compiler
      itself inserts between the MFMA and a dependent v_add are counted);
  R2  a VGPR written by a VALU instruction must not be read as the DPP source of the next two wait states
      (VALU write -> DPP read: 2), nor by v_permlane*_swap (2);
  R3  EXEC written by a VALU instruction (v_cmpx*) must not be followed by a DPP instruction within 5 wait states (EXEC
      written by the scalar unit -- s_and_saveexec, s_mov exec -- is interlocked);
  R4  a VGPR written by v_dot2* must not be read by a VALU within 3 wait states;
  R5  an SGPR written by a VALU (v_readlane / v_readfirstlane / v_cmp) must not be used as the lane select of
      v_readlane / v_writelane within 4 wait states.

A wait state is one issued instruction; `s_nop N` counts N + 1.  The walk is linear (fall-through order): it does not follow
back-edges, so a hazard that exists only across a loop's back-edge is not seen -- a lint, not a proof.  Independent
instructions between producer and consumer count, exactly as the hardware counts them.

Usage:  python tools/isa_hazard_lint.py keyword-spotting_amd/csrc/kws_dscnn.hip [-DKWS_X_...]   (exit code 1 on findings)
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "keyword-spotting_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-ffp-contract=off",
              "-DKWS_BUILD", "--offload-device-only", "-S", "-Wno-unused-command-line-argument"]
CACHE = os.path.join(tempfile.gettempdir(), "kws_isa_lint")

VALU_DPP_WAIT = 2
VALU_MFMA_WAIT = 2          # VALU write -> MFMA A/B/C operand read of the same VGPR
EXEC_DPP_WAIT = 5
DOT_VALU_WAIT = 3
VALU_SGPR_LANESEL_WAIT = 4


def compile_to_isa(src: str, extra=()) -> str:
    """Path of the ISA listing of `src` (cached on the contents of the source, the headers next to it and the flags)."""
    os.makedirs(CACHE, exist_ok=True)
    h = hashlib.sha256()
    d = os.path.dirname(os.path.abspath(src))
    for f in sorted(os.listdir(d)) + [os.path.join(ROOT, "include", "kws_hip.h")]:
        path = f if os.path.isabs(f) else os.path.join(d, f)
        if path.endswith((".hip", ".h")) and os.path.isfile(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    h.update(open(src, "rb").read())
    h.update(" ".join(extra).encode())
    out = os.path.join(CACHE, f"{os.path.basename(src)}.{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        subprocess.run([HIPCC, *BASE_FLAGS, *extra, src, "-o", out], check=True, capture_output=True, text=True)
    return out


# ------------------------------------------------------------------------------------------------ parsing
_REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")


def regs_of(operand: str):
    """Set of (bank, index) named by one operand: v5, a[0:15], s[4:5] ...; modifiers like |v1|, -v2, neg(...) included."""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("line", "text", "op", "operands", "waits")

    def __init__(self, line, text):
        self.line, self.text = line, text
        parts = text.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands end where the modifiers begin (dpp controls, op_sel, offset: ...)
        self.operands = [o.strip() for o in rest.split(",")]
        self.waits = (int(rest.strip(), 0) + 1) if self.op == "s_nop" else 1


def functions(isa_path: str):
    """{name: [Ins]} for every function (kernel or device function) of the listing."""
    funcs, cur, name = {}, None, None
    with open(isa_path) as f:
        for n, raw in enumerate(f, 1):
            line = raw.split(";", 1)[0].rstrip()
            if not line:
                continue
            if not line.startswith(("\t", " ")):
                m = re.match(r"^([A-Za-z_][\w$.]*):", line)
                if m and not m.group(1).startswith((".L", "__hip_cuid")):
                    name, cur = m.group(1), []
                    funcs[name] = cur
                continue
            t = line.strip()
            if t.startswith("."):
                if t.startswith((".end_amdhsa_kernel", ".section", ".text")) and cur is not None and t.startswith(".section"):
                    cur = None
                continue
            if cur is not None and re.match(r"^[a-z]", t):
                cur.append(Ins(n, t))
    return {k: v for k, v in funcs.items() if v}


def is_mfma(op):
    return op.startswith(("v_mfma", "v_smfmac"))


def is_valu(op):
    return op.startswith("v_") and not is_mfma(op)


def is_dpp(ins):
    return ins.op.endswith("_dpp") or "row_shr" in ins.text or "row_shl" in ins.text or "row_bcast" in ins.text or \
        "wave_shr" in ins.text or "wave_shl" in ins.text or "quad_perm" in ins.text or "row_ror" in ins.text or \
        "row_mirror" in ins.text or "row_half_mirror" in ins.text or "row_newbcast" in ins.text


def reads_vgpr_data(op):
    """Instructions whose VGPR sources are read by the VALU / LDS / memory pipes (R1's consumers)."""
    return is_valu(op) or op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_"))


def split_dst_src(ins: Ins):
    """(written registers, read registers).  VALU / MFMA: first operand is the destination (v_cmp*: VCC or an SGPR pair;
    v_readlane: an SGPR); stores and most LDS writes have no VGPR destination."""
    op, ops = ins.op, ins.operands
    if not ops or not ops[0]:
        return set(), set()
    clean = [re.split(r"\s+(?=[a-z_]+:|row_|wave_|quad_|bank_|bound_|op_sel|neg_|clamp|mul:|div:|offset|glc|slc|sc0|sc1|nt|gds)", o)[0] for o in ops]
    if op.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store")) or op.startswith("ds_bpermute") is False and op.startswith("ds_") and "write" in op:
        return set(), set().union(*(regs_of(o) for o in clean))
    if op.startswith("v_permlane") and "swap" in op:  # both operands are read and written
        both = set().union(*(regs_of(o) for o in clean))
        return both, both
    dst = regs_of(clean[0])
    src = set().union(*(regs_of(o) for o in clean[1:])) if len(clean) > 1 else set()
    if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c", "v_pk_fmac")):
        src |= dst  # accumulating forms read their destination
    return dst, src


# ------------------------------------------------------------------------------------------------ calibration
_CALIB_SRC = r"""
#include <hip/hip_runtime.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
extern "C" __global__ void calib_f16(const halfx8* a, const halfx8* b, float* o) {
    floatx16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    o[threadIdx.x] = c[0] + c[5] * 3.0f;
}
extern "C" __global__ void calib_bf16(const bf16x8* a, const bf16x8* b, float* o) {
    floatx16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    o[threadIdx.x] = c[0] + c[5] * 3.0f;
}
extern "C" __global__ void calib_f32(const float* a, const float* b, float* o) {
    floatx16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    o[threadIdx.x] = c[0] + c[5] * 3.0f;
}
"""


def calibrate_mfma_waits():
    """{mfma opcode: wait states the compiler itself leaves between the MFMA and a VALU read of its result}."""
    os.makedirs(CACHE, exist_ok=True)
    src = os.path.join(CACHE, "calib.hip")
    with open(src, "w") as f:
        f.write(_CALIB_SRC)
    isa = compile_to_isa(src)
    out = {}
    for name, body in functions(isa).items():
        for i, ins in enumerate(body):
            if not is_mfma(ins.op):
                continue
            dst, _ = split_dst_src(ins)
            waits = 0
            for nxt in body[i + 1:]:
                _, src_regs = split_dst_src(nxt)
                if reads_vgpr_data(nxt.op) and (src_regs & dst):
                    out[ins.op] = min(out.get(ins.op, 1 << 30), waits)
                    break
                waits += nxt.waits
    return out


# ------------------------------------------------------------------------------------------------ the walk
def lint_function(body, mfma_waits, default_mfma_wait):
    """Findings [(line, rule, message)] of one function."""
    findings = []
    mfma_age = {}     # vgpr -> (wait states since the MFMA that wrote it issued, required, opcode, line)
    valu_age = {}     # vgpr -> wait states since a (non-DPP-exempt) VALU wrote it
    dot_age = {}      # vgpr -> wait states since v_dot2* wrote it
    sgpr_valu_age = {}
    exec_age = None
    for ins in body:
        op = ins.op
        dst, src = split_dst_src(ins)
        # ---- checks (the state describes what was issued BEFORE this instruction)
        if reads_vgpr_data(op):
            for r in src:
                if r in mfma_age:
                    age, need, mop, mline = mfma_age[r]
                    if age < need:
                        findings.append((ins.line, "R1", f"`{ins.text}` reads {r[0]}{r[1]} {age} wait state(s) after `{mop}` (line {mline}) wrote it; needs {need}"))
        if is_mfma(op):
            # R6: a VGPR written by a VALU instruction (possibly inside an asm block the compiler cannot see into) and read as an
            # MFMA operand needs VALU_MFMA_WAIT wait states in between (cdna_hip_programming.md 5.7 item 2: `s_nop 1` after a
            # just-written "v" operand; the compiler pads only one state after ;;#ASMEND).  MFMA -> MFMA dependencies (srcC
            # forwarding) are scheduled by the compiler; no asm block issues MFMAs.
            for r in src:
                if r in valu_age and valu_age[r] < VALU_MFMA_WAIT:
                    findings.append((ins.line, "R6", f"`{ins.text}` reads {r[0]}{r[1]} as a matrix operand {valu_age[r]} wait state(s) after a VALU wrote it; needs {VALU_MFMA_WAIT}"))
        if is_dpp(ins) or (op.startswith("v_permlane") and "swap" in op):
            # DPP source = src0 (first source operand); permlane swaps read both operands
            srcs = src if op.startswith("v_permlane") else (regs_of(ins.operands[1]) if len(ins.operands) > 1 else set())
            for r in srcs:
                if r in valu_age and valu_age[r] < VALU_DPP_WAIT:
                    findings.append((ins.line, "R2", f"`{ins.text}` reads {r[0]}{r[1]} through DPP / permlane {valu_age[r]} wait state(s) after a VALU wrote it; needs {VALU_DPP_WAIT}"))
            if is_dpp(ins) and exec_age is not None and exec_age < EXEC_DPP_WAIT:
                findings.append((ins.line, "R3", f"`{ins.text}` is a DPP instruction {exec_age} wait state(s) after EXEC was written; needs {EXEC_DPP_WAIT}"))
        if is_valu(op):
            for r in src:
                if r in dot_age and dot_age[r] < DOT_VALU_WAIT:
                    findings.append((ins.line, "R4", f"`{ins.text}` reads {r[0]}{r[1]} {dot_age[r]} wait state(s) after v_dot2 wrote it; needs {DOT_VALU_WAIT}"))
        if op.startswith(("v_readlane", "v_writelane")) and len(ins.operands) >= 3:
            for r in regs_of(ins.operands[2]):
                if r[0] == "s" and r in sgpr_valu_age and sgpr_valu_age[r] < VALU_SGPR_LANESEL_WAIT:
                    findings.append((ins.line, "R5", f"`{ins.text}` uses s{r[1]} as lane select {sgpr_valu_age[r]} wait state(s) after a VALU wrote it; needs {VALU_SGPR_LANESEL_WAIT}"))
        # ---- age everything by this instruction's wait states, then record its writes
        w = ins.waits
        for d in (mfma_age,):
            for k in list(d):
                a, need, mop, ml = d[k]
                if a + w >= 64:
                    del d[k]
                else:
                    d[k] = (a + w, need, mop, ml)
        for d in (valu_age, dot_age, sgpr_valu_age):
            for k in list(d):
                if d[k] + w >= 16:
                    del d[k]
                else:
                    d[k] += w
        if exec_age is not None:
            exec_age = exec_age + w if exec_age + w < 16 else None
        if is_mfma(op):
            need = mfma_waits.get(op, default_mfma_wait)
            for r in dst:
                mfma_age[r] = (0, need, op, ins.line)
                valu_age.pop(r, None)
        elif is_valu(op):
            for r in dst:
                if r[0] in ("v", "a"):
                    valu_age[r] = 0
                    mfma_age.pop(r, None)
                    dot_age.pop(r, None)
                    if op.startswith("v_dot2"):
                        dot_age[r] = 0
                elif r[0] == "s":
                    sgpr_valu_age[r] = 0
            if op.startswith("v_cmpx"):
                exec_age = 0
        elif op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # no fall-through: what follows in the text is reached only through a label, from code this linear walk has not
            # just seen (the compiler schedules those joins; the hand-written asm blocks this lint is for are straight-line)
            mfma_age.clear()
            valu_age.clear()
            dot_age.clear()
            sgpr_valu_age.clear()
            exec_age = None
        elif op.startswith(("ds_", "global_load", "buffer_load", "flat_load", "scratch_load")):
            for r in dst:  # a load's destination is rewritten later, behind s_waitcnt: the old producers no longer matter
                mfma_age.pop(r, None)
                valu_age.pop(r, None)
                dot_age.pop(r, None)
    return findings


def lint_file(src: str, extra=()):
    """(findings per function, calibrated MFMA waits) for one .hip translation unit."""
    waits = calibrate_mfma_waits()
    default = max(waits.values()) if waits else 19
    isa = compile_to_isa(src, extra)
    res = {}
    for name, body in functions(isa).items():
        f = lint_function(body, waits, default)
        if f:
            res[name] = f
    return res, waits, isa


def main(argv):
    if not argv:
        print(__doc__)
        return 2
    src, extra = argv[0], argv[1:]
    res, waits, isa = lint_file(src, extra)
    print(f"{src}: ISA {isa}; calibrated MFMA -> VALU waits {waits}")
    n = 0
    for name, fs in res.items():
        print(f"  {name}: {len(fs)} finding(s)")
        for line, rule, msg in fs[:8]:
            print(f"    line {line} [{rule}] {msg}")
        n += len(fs)
    print("clean" if n == 0 else f"{n} finding(s)")
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
