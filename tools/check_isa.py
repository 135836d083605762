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


class IsaCheckError(RuntimeError):
    pass


def device_disassembly(obj, arch):
    """Disassembly of the object's device code for `arch`; "" only when the object has no .hip_fatbin section at all (a host-only
    object). Anything else that keeps the check from LOOKING - an unbundle failure (renamed bundle id, another --offload-arch), an
    empty disassembly - is an error: the guard fails closed (advisor r4)."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "a.fatbin"), os.path.join(td, "a.co")
        sect = subprocess.run([f"{LLVM}/llvm-readelf", "-S", obj], capture_output=True, text=True)
        if sect.returncode:
            raise IsaCheckError(f"{obj}: cannot read section headers: {sect.stderr.strip()}")
        if ".hip_fatbin" not in sect.stdout:
            return ""                                   # no device code in this object
        if subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat]).returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            raise IsaCheckError(f"{obj}: has a .hip_fatbin section but objcopy could not extract it")
        target = f"hipv4-amdgcn-amd-amdhsa--{arch}"
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={target}", f"--input={fat}",
                            f"--output={co}", "--unbundle"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            ids = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--input={fat}", "--list"], capture_output=True, text=True).stdout.split()
            raise IsaCheckError(f"{obj}: no code object for {target} in its fat binary (bundles: {ids}); pass the Makefile's ARCH (--arch)")
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True)
        if dis.returncode or not re.search(r"^\s+[sv]_\w+", dis.stdout, re.M):
            raise IsaCheckError(f"{obj}: llvm-objdump produced no instructions for the {target} code object")
        return dis.stdout


def main(argv):
    arch = "gfx950"
    objs = []
    it = iter(argv)
    for a in it:
        if a == "--arch":
            arch = next(it)
        elif a.startswith("--arch="):
            arch = a.split("=", 1)[1]
        else:
            objs.append(a)
    bad = 0
    looked = 0
    for obj in objs:
        try:
            dis = device_disassembly(obj, arch)
        except IsaCheckError as e:
            print(f"check_isa: FAILED - {e}", file=sys.stderr)
            return 2
        if not dis:
            continue
        looked += 1
        hits = [ln.split("//")[0].strip() for ln in dis.splitlines() if PAT.search(ln)]
        packed = sum(1 for ln in dis.splitlines() if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln))
        if hits:
            bad += len(hits)
            print(f"check_isa: {os.path.basename(obj)}: {len(hits)} packed-f32 instruction(s) with a crossed op_sel, e.g. `{hits[0]}`", file=sys.stderr)
        elif os.environ.get("CHECK_ISA_VERBOSE"):
            print(f"check_isa: {os.path.basename(obj)}: {packed} packed-f32 instructions, none with op_sel")
    if objs and not looked:
        print("check_isa: FAILED - none of the objects carries device code: nothing was checked", file=sys.stderr)
        return 2
    if bad:
        print("check_isa: FAILED - build the object with -fno-slp-vectorize or rewrite the expression (DESIGN.md section 5d)", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
