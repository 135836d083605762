"""ISA-level bisect of the sampling-kernel defect: builds code objects of `msda_fused_kernel<f16,1,2>` from the
compiler's OWN assembly, patched instruction by instruction, for `tools/msda_isa_probe.py` to run beside the other chain's MFMA kernels.

    python tools/msda_isa_variants.py            # -> tools/_msda_isa/<variant>.hsaco + variants.json   (CPU only: hipcc, clang, ld.lld)

`hipcc -S` of csrc/msda.hip with SLP vectorisation (the failing build) and without (the shipped build) gives two assembly files. Every
other variant is the SLP assembly with a text patch inside the one kernel: `s_nop`s around instruction classes, or packed-f32
instructions (`v_pk_{fma,mul,add}_f32`) rewritten into the two scalar VALU instructions that compute the same halves (same rounding:
the outputs stay bit-identical, which the probe checks), for chosen subsets of the packed instructions.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_msda_isa")
KERNEL = "_ZN12_GLOBAL__N_117msda_fused_kernelIDF16_Li1ELi2EEEvNS_10MsdaParamsE"
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/lw-detr_amd/csrc"]
TMP = (57, 58)            # v57 / v58: allocated (granule of 8, accum_offset 60) but unused by the kernel (next_free_vgpr 57)


def compile_asm(extra, path):
    subprocess.check_call(["hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", os.path.join(ROOT, "lw-detr_amd/csrc/msda.hip"), "-o", path],
                          stderr=subprocess.DEVNULL)


def assemble(asm_path, hsaco_path):
    obj = hsaco_path + ".o"
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm_path, "-o", obj])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", obj, "-o", hsaco_path])
    os.remove(obj)


def kernel_span(lines):
    a = next(i for i, ln in enumerate(lines) if ln.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return a, b


# ---------------------------------------------------------------------------------- packed f32 -> two scalar instructions
def _split_ops(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def _half(op, sel):
    """operand text, half selector (0 = low register of the pair, 1 = high) -> scalar operand text"""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + sel}"
    return op                     # inline constant: the same value for both halves (the compiler sets op_sel_hi 0 for it)


def _regs(op):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    return (int(m.group(1)), int(m.group(2))) if m else None


def scalarize(line):
    """One v_pk_{fma,mul,add}_f32 line -> list of scalar VALU lines computing the same two halves (temporaries where halves cross)."""
    body = line.split(";")[0].strip()
    m = re.match(r"(v_pk_(fma|mul|add)_f32)\s+(.*)", body)
    kind, rest = m.group(2), m.group(3)
    mods = {k: [int(v) for v in vals.split(",")] for k, vals in re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([0-9,]+)\]", rest)}
    rest = re.sub(r"\s*(op_sel_hi|op_sel|neg_lo|neg_hi):\[[0-9,]+\]", "", rest).strip()
    ops = _split_ops(rest)
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    assert n == (3 if kind == "fma" else 2), line
    sel_lo, sel_hi = mods.get("op_sel", [0] * n), mods.get("op_sel_hi", [1] * n)
    neg_lo, neg_hi = mods.get("neg_lo", [0] * n), mods.get("neg_hi", [0] * n)
    d0 = _regs(dst)[0]

    def half_ops(sel, neg):
        res = []
        for i, s_ in enumerate(srcs):
            t = _half(s_, sel[i])
            if neg[i]:
                t = ("-" + t) if not t.startswith("-") else t[1:]
            res.append(t)
        return res

    lo, hi = half_ops(sel_lo, neg_lo), half_ops(sel_hi, neg_hi)
    mn = {"fma": "v_fma_f32", "mul": "v_mul_f32_e64", "add": "v_add_f32_e64"}[kind]
    reads = lambda lst: {t.lstrip("-") for t in lst}
    out = []
    if f"v{d0}" not in reads(hi):                       # lo first is safe
        out = [f"\t{mn} v{d0}, {', '.join(lo)}", f"\t{mn} v{d0 + 1}, {', '.join(hi)}"]
    elif f"v{d0 + 1}" not in reads(lo):                 # hi first is safe
        out = [f"\t{mn} v{d0 + 1}, {', '.join(hi)}", f"\t{mn} v{d0}, {', '.join(lo)}"]
    else:                                               # halves cross: lo into a temporary
        out = [f"\t{mn} v{TMP[0]}, {', '.join(lo)}", f"\t{mn} v{d0 + 1}, {', '.join(hi)}", f"\tv_mov_b32_e32 v{d0}, v{TMP[0]}"]
    return out


def is_pk(ln):
    return re.match(r"\s*v_pk_(fma|mul|add)_f32\s", ln) is not None


def classify(lines, a, b):
    """Packed instructions of the kernel, in order: (line index, class). Classes: 'acc' = the bilinear accumulation
    (v_pk_fma_f32 whose first source broadcasts one register: op_sel_hi:[0,1,*]), 'loc' = everything before it (location arithmetic)."""
    res = []
    for i in range(a, b):
        if is_pk(lines[i]):
            body = lines[i].split(";")[0]
            acc = "v_pk_fma_f32" in body and re.search(r"op_sel_hi:\[0,1,[01]\]", body) is not None and "op_sel:" not in body
            res.append((i, "acc" if acc else "loc"))
    return res


def patch(lines, a, b, *, scalar=lambda idx, cls, ln: False, nop_before=None, nop_after=None, nops=1):
    """-> new line list. scalar(k, cls, line): rewrite the k-th packed instruction; nop_before / nop_after: regex of instructions."""
    pk = {i: (k, cls) for k, (i, cls) in enumerate(classify(lines, a, b))}
    out = list(lines[:a])
    for i in range(a, b):
        ln = lines[i]
        body = ln.split(";")[0]
        if nop_before and re.match(nop_before, body.strip()):
            out.append(f"\ts_nop {nops - 1}")
        if i in pk and scalar(pk[i][0], pk[i][1], ln):
            out.extend(scalarize(ln))
        else:
            out.append(ln)
        if nop_after and re.match(nop_after, body.strip()):
            out.append(f"\ts_nop {nops - 1}")
    out.extend(lines[b:])
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    slp_s, noslp_s = os.path.join(OUT, "slp.s"), os.path.join(OUT, "noslp.s")
    compile_asm([], slp_s)
    compile_asm(["-fno-slp-vectorize"], noslp_s)
    lines = open(slp_s).read().split("\n")
    a, b = kernel_span(lines)
    pk = classify(lines, a, b)
    n_loc, n_acc = sum(c == "loc" for _, c in pk), sum(c == "acc" for _, c in pk)
    # accumulation instructions whose packed operand pair was written by the instruction right in front (cvt -> pk back to back)
    def fed_by_prev(i):
        prev = lines[i - 1].split(";")[0]
        m = re.match(r"\s*v_cvt_f32_f16\w*\s+v(\d+)", prev)
        srcs = re.findall(r"v\[(\d+):(\d+)\]", lines[i].split(";")[0])[1:]
        return bool(m) and any(int(lo) <= int(m.group(1)) <= int(hi) for lo, hi in srcs)
    b2b = {k for k, (i, c) in enumerate(pk) if c == "acc" and fed_by_prev(i)}
    acc_idx = [k for k, (_, c) in enumerate(pk) if c == "acc"]
    variants = {
        "slp": ("compiler output with SLP vectorisation (the failing build)", None),
        "noslp": ("compiler output with -fno-slp-vectorize (the shipped build)", None),
        "all_scalar": ("every packed-f32 instruction rewritten as two scalar ones (checks the rewriter: must behave as noslp)", dict(scalar=lambda k, c, l: True)),
        "loc_scalar": (f"the {n_loc} packed instructions of the location arithmetic scalar, the {n_acc} of the accumulation packed", dict(scalar=lambda k, c, l: c == "loc")),
        "acc_scalar": (f"the {n_acc} packed FMAs of the accumulation scalar, the location arithmetic packed", dict(scalar=lambda k, c, l: c == "acc")),
        "acc_first_half_scalar": ("first half of the accumulation FMAs scalar", dict(scalar=lambda k, c, l: c == "loc" or k in acc_idx[:len(acc_idx) // 2])),
        "acc_second_half_scalar": ("second half of the accumulation FMAs scalar", dict(scalar=lambda k, c, l: c == "loc" or k in acc_idx[len(acc_idx) // 2:])),
        "acc_b2b_scalar": (f"location arithmetic scalar + the {len(b2b)} accumulation FMAs that read a register written by the v_cvt right in front of them scalar",
                           dict(scalar=lambda k, c, l: c == "loc" or k in b2b)),
        "acc_not_b2b_scalar": ("location arithmetic scalar + the accumulation FMAs NOT fed by the preceding instruction scalar", dict(scalar=lambda k, c, l: c == "loc" or (c == "acc" and k not in b2b))),
        "nop1_before_pk": ("s_nop 0 in front of every packed-f32 instruction", dict(nop_before=r"v_pk_(fma|mul|add)_f32", nops=1)),
        "nop2_before_pk": ("s_nop 1 in front of every packed-f32 instruction", dict(nop_before=r"v_pk_(fma|mul|add)_f32", nops=2)),
        "nop4_before_pk": ("s_nop 3 in front of every packed-f32 instruction", dict(nop_before=r"v_pk_(fma|mul|add)_f32", nops=4)),
        "nop2_after_pk": ("s_nop 1 behind every packed-f32 instruction", dict(nop_after=r"v_pk_(fma|mul|add)_f32", nops=2)),
        "nop2_after_cvt": ("s_nop 1 behind every v_cvt_f32_f16 (plain and SDWA)", dict(nop_after=r"v_cvt_f32_f16", nops=2)),
        "nop2_after_cndmask": ("s_nop 1 behind every v_cndmask_b32", dict(nop_after=r"v_cndmask_b32", nops=2)),
        "nop8_before_pk": ("s_nop 7 in front of every packed-f32 instruction", dict(nop_before=r"v_pk_(fma|mul|add)_f32", nops=8)),
    }
    loc_idx = [k for k, (_, c) in enumerate(pk) if c == "loc"]
    for n_, k_ in enumerate(loc_idx):
        txt = " ".join(lines[pk[k_][0]].split(";")[0].split())
        variants[f"only_loc{n_:02d}"] = (f"ONLY this instruction packed, all others scalar: {txt}", dict(scalar=lambda k, c, l, k_=k_: k != k_))
    crossed = [k for k in loc_idx if "op_sel:" in lines[pk[k][0]]]
    sgpr = [k for k in loc_idx if re.search(r"\bs\[\d+:\d+\]", lines[pk[k][0]].split(";")[0])]
    variants["only_crossed"] = (f"only the {len(crossed)} packed instructions with a crossed op_sel (a half reads the OTHER register of a pair) packed", dict(scalar=lambda k, c, l: k not in crossed))
    variants["only_sgpr"] = (f"only the {len(sgpr)} packed instructions with an SGPR-pair source packed", dict(scalar=lambda k, c, l: k not in sgpr))
    variants["loc_but_crossed"] = ("location arithmetic packed except the crossed-op_sel instructions, accumulation packed", dict(scalar=lambda k, c, l: k in crossed))
    variants["loc_but_sgpr"] = ("location arithmetic packed except the SGPR-pair instructions, accumulation packed", dict(scalar=lambda k, c, l: k in sgpr))
    meta = {"kernel": KERNEL, "packed_total": len(pk), "packed_loc": n_loc, "packed_acc": n_acc, "acc_fed_by_previous_cvt": len(b2b), "variants": {}}
    for name, (desc, kw) in variants.items():
        src = os.path.join(OUT, name + ".s")
        if kw is None:
            src = slp_s if name == "slp" else noslp_s
        else:
            open(src, "w").write("\n".join(patch(lines, a, b, **kw)))
        assemble(src, os.path.join(OUT, name + ".hsaco"))
        body = open(src).read().split("\n")
        ka, kb = kernel_span(body)
        meta["variants"][name] = {"desc": desc, "pk_left": sum(is_pk(l) for l in body[ka:kb]), "instructions": sum(1 for l in body[ka:kb] if l.startswith("\t") and not l.strip().startswith((".", ";")))}
        print(f"{name:26s} packed left {meta['variants'][name]['pk_left']:3d}  instructions {meta['variants'][name]['instructions']:4d}  {desc}")
    json.dump(meta, open(os.path.join(OUT, "variants.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
