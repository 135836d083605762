#!/bin/bash
# round 3, GPU session 14: block-kernel grid = needed workgroups (default now) vs whole rounds; two launch chains on two streams
set -u
OUT=gpurun_out/r3_s14
mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-latency --no-roofline --steps 30 --warmup 5 $BARGS > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - "$tag" $OUT/bench_$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1]); print(tag, d["value"], d["ms_per_step"])
except Exception as e:
    print(tag, "FAILED", e, open(path.replace(".json", ".err")).read()[-600:])
PY
}
BARGS="--size small --batch 32 --dtype fp16"
run small_default X=1
run small_rounds LWDETR_VB_GRID=rounds
run small_2streams LWDETR_STREAMS=2
run small_2streams_rounds LWDETR_STREAMS=2 LWDETR_VB_GRID=rounds
BARGS="--size medium --batch 64 --dtype bf16"
run medium_default X=1
run medium_rounds LWDETR_VB_GRID=rounds
run medium_2streams LWDETR_STREAMS=2
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vit_block" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_msda.py -x -q -m gpu -k "gradcheck or large_channel" 2>&1 | tail -2
timeout 200 python - <<'PY'
# two-stream forward == one-stream forward, bit for bit (same kernels on half batches; images independent)
import os, torch, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
import lwdetr_amd
from lwdetr_amd.synth import synth_images, synth_state_dict
m, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args("small"))
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0)); m = m.cuda().half().eval()
x = synth_images(16, 640, 640, seed=7).cuda().half()
o = m(x); torch.cuda.synchronize()
torch.save({k: o[k].cpu() for k in ("pred_logits", "pred_boxes")}, sys.argv[1])
'''
for s_ in ("1", "2"):
    subprocess.check_call([sys.executable, "-c", code, f"/tmp/o{s_}.pt"], env=dict(os.environ, LWDETR_STREAMS=s_))
a, b = torch.load("/tmp/o1.pt"), torch.load("/tmp/o2.pt")
print("two-stream vs one-stream (B=16): logits equal", torch.equal(a["pred_logits"], b["pred_logits"]), "boxes equal", torch.equal(a["pred_boxes"], b["pred_boxes"]),
      "max|d|", (a["pred_logits"].float() - b["pred_logits"].float()).abs().max().item())
PY
