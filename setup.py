"""Optional packaging: `pip install --no-build-isolation .` installs the in-tree package directory `lw-detr_amd/` under its
import name `lwdetr_amd` together with the prebuilt `liblwdetr_hip.so` (build it first: `make -C lw-detr_amd/csrc -j`, or
`python -c 'import __graft_entry__ as g; g.build()'`). Working from a checkout needs none of this: `lwdetr_amd.py` at the
repository root aliases the directory."""
import os

from setuptools import setup

HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(HERE, "lw-detr_amd", "liblwdetr_hip.so")):
    raise SystemExit("build lw-detr_amd/liblwdetr_hip.so first (make -C lw-detr_amd/csrc -j): the wheel ships the HIP library")

setup(
    name="lwdetr-amd",
    version="0.2.0",
    description="MI355X-native (gfx950) LW-DETR inference forward path: hand-written HIP kernels behind the reference's API",
    packages=["lwdetr_amd", "lwdetr_amd.models", "lwdetr_amd.ops"],
    package_dir={"lwdetr_amd": "lw-detr_amd"},
    package_data={"lwdetr_amd": ["liblwdetr_hip.so", "compat/*.py"]},      # compat/: sys.path shim dir
    python_requires=">=3.9",
)
