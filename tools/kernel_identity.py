"""Are the device kernels of two builds of an object instruction-identical? (round 6: pruning the opt-in experiment kernels out of the default
build must not move the product kernels - the round-5 lesson, profiles/r5g_*: a never-taken branch in a shared epilogue cost every large-tile
GEMM 15-20 % through register allocation.)

    python tools/kernel_identity.py before/gemm.o after/gemm.o [--arch gfx950]
Prints, per kernel symbol: identical / DIFFERENT (instruction counts) / only in one of the two. Exit code 1 if a kernel present in both differs."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import device_disassembly  # noqa: E402


def kernels(obj, arch):
    out, name, body = {}, None, []
    for line in device_disassembly(obj, arch).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                out[name] = body
            name, body = m.group(1), []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*[0-9A-Fa-f]+:", line)         # "  s_load_dwordx2 s[0:1], ...   // 000000001000: C0060002 ..."
        if m and name:
            ins = re.sub(r"\s+", " ", m.group(1))
            # pc-relative distance to a global symbol (s_getpc_b64 + s_add_u32 / s_addc_u32 of a literal): moves with the kernel's place in the object
            ins = re.sub(r"^(s_addc?_u32 s\d+, s\d+, )0x[0-9a-f]{4,}$", r"\1<pcrel>", ins)
            body.append(ins)
    if name:
        out[name] = body
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    arch = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--arch=")), "gfx950")
    a, b = kernels(args[0], arch), kernels(args[1], arch)
    bad = 0
    for k in sorted(set(a) | set(b)):
        short = k if len(k) < 110 else k[:107] + "..."
        if k not in a or k not in b:
            print(f"only in {'first' if k in a else 'second'}: {short}")
        elif a[k] == b[k]:
            print(f"identical ({len(a[k])} instructions): {short}")
        else:
            bad += 1
            print(f"DIFFERENT ({len(a[k])} vs {len(b[k])} instructions): {short}")
    print(f"{bad} kernel(s) present in both builds differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
