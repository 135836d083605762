"""CPU: the build-time ISA guard (tools/check_isa.py, run by lw-detr_amd/csrc/Makefile on every object) - no packed-f32 VALU
instruction with a non-default op_sel in the product's device code (`v_pk_fma_f32 ... op_sel:[0,1,0]
op_sel_hi:[1,0,0]` returns wrong lanes beside another kernel's MFMA waves on MI355X)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tools", "check_isa.py")


def test_product_objects_carry_no_crossed_packed_f32():
    objs = sorted(glob.glob(os.path.join(ROOT, "lw-detr_amd", "csrc", "build", "*.o")))
    if not objs:
        pytest.skip("library objects not built here")
    r = subprocess.run([sys.executable, CHECK] + objs, capture_output=True, text=True, env=dict(os.environ, CHECK_ISA_VERBOSE="1"))
    assert r.returncode == 0, r.stderr
    # the check really looked at device code: the GEMM / attention objects hold thousands of (straight) packed-f32 instructions
    counts = {ln.split(":")[1].strip(): int(ln.split(":")[2].split()[0]) for ln in r.stdout.splitlines() if "packed-f32 instructions" in ln}
    assert counts.get("gemm.o", 0) > 1000 and counts.get("msda.o", 1) == 0 and counts.get("topk.o", 1) == 0, counts


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_guard_flags_the_crossed_form(tmp_path):
    src = tmp_path / "bad.hip"
    src.write_text('''#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o, f2 x, f2 b) {
    f2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel:[0,1,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(b));
    o[threadIdx.x] = r[0] + r[1];
}
''')
    obj = tmp_path / "bad.o"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, CHECK, str(obj)], capture_output=True, text=True)
    assert r.returncode == 1 and "crossed op_sel" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_guard_fails_closed(tmp_path):
    """An object whose device code the guard cannot look at is an ERROR, not a pass (advisor r4): another --offload-arch than the
    one the Makefile passes, or only host objects on the command line."""
    src = tmp_path / "ok.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float* o) { o[threadIdx.x] = 1.f; }\n")
    obj = tmp_path / "ok.o"
    subprocess.check_call(["hipcc", "--offload-arch=gfx942", "-O3", "-c", str(src), "-o", str(obj)], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, CHECK, "--arch", "gfx950", str(obj)], capture_output=True, text=True)
    assert r.returncode == 2 and "no code object for" in r.stderr, (r.returncode, r.stderr)
    assert subprocess.run([sys.executable, CHECK, "--arch", "gfx942", str(obj)], capture_output=True, text=True).returncode == 0
    host = tmp_path / "host.o"
    (tmp_path / "host.c").write_text("int f(void) { return 1; }\n")
    subprocess.check_call(["gcc", "-c", str(tmp_path / "host.c"), "-o", str(host)])
    r = subprocess.run([sys.executable, CHECK, str(host)], capture_output=True, text=True)
    assert r.returncode == 2 and "nothing was checked" in r.stderr, (r.returncode, r.stderr)
