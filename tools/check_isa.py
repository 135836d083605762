"""Build-time guard (csrc/Makefile runs it on every object): no packed-f32 VALU instruction with a non-default `op_sel` may be in the
product's device code.

Why (profiles/r4a_msda_isa_bisect.txt): on MI355X `v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,0,0]` - the low
result reading the HIGH register of a source pair and vice versa - returned wrong values in lanes 48-63 of a few waves per launch
whenever MFMA waves of another kernel shared the SIMD (two launch chains); the straight forms (default op_sel, op_sel_hi only), which
hipcc emits tens of thousands of times in the GEMM / attention / LayerNorm epilogues, never did. hipcc's SLP vectoriser creates the
crossed form when it pairs two scalar operations whose operands sit in swapped order; msda.o and topk.o are therefore built with
-fno-slp-vectorize, and this check fails the build if any object carries such an instruction again (new compiler, new code)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PAT = re.compile(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[")


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "a.fatbin"), os.path.join(td, "a.co")
        if subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat]).returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""                                   # no device code in this object
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                            f"--output={co}", "--unbundle"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co):
            return ""
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout


def main(objs):
    bad = 0
    for obj in objs:
        dis = device_disassembly(obj)
        hits = [ln.split("//")[0].strip() for ln in dis.splitlines() if PAT.search(ln)]
        packed = sum(1 for ln in dis.splitlines() if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln))
        if hits:
            bad += len(hits)
            print(f"check_isa: {os.path.basename(obj)}: {len(hits)} packed-f32 instruction(s) with a crossed op_sel, e.g. `{hits[0]}`", file=sys.stderr)
        elif os.environ.get("CHECK_ISA_VERBOSE"):
            print(f"check_isa: {os.path.basename(obj)}: {packed} packed-f32 instructions, none with op_sel")
    if bad:
        print("check_isa: FAILED - build the object with -fno-slp-vectorize or rewrite the expression (DESIGN.md section 5d)", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
