"""Register / LDS / scratch usage of every kernel in an object (from the code object's metadata note):
    python tools/kernel_regs.py lw-detr_amd/csrc/build/gemm.o [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main(obj, flt=""):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "a.fatbin"), os.path.join(td, "a.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        sym = subprocess.run(["c++filt"], input=notes, capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for ln in sym.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count" and cur.get("name"):
            rows.append(cur); cur = {}
        cur[k] = v
    if cur.get("name"):
        rows.append(cur)
    for r in rows:
        name = r.get("name", "?")
        if flt and flt not in name:
            continue
        print(f"vgpr {r.get('vgpr_count', '?'):>4} agpr {r.get('agpr_count', '?'):>4} sgpr {r.get('sgpr_count', '?'):>4} spill {r.get('vgpr_spill_count', '?'):>4} "
              f"scratch {r.get('private_segment_fixed_size', '?'):>5} lds {r.get('group_segment_fixed_size', '?'):>6}  {name[:150]}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
