#!/bin/bash
# round 3, GPU session 48: top-k with four histogram copies: tests, per-launch times of the selection / PostProcess kernels
set -u
timeout 900 python -m pytest tests/test_gpu_topk.py -x -q -m gpu 2>&1 | tail -1
python - <<'PY'
import torch, lwdetr_amd
from lwdetr_amd.synth import synth_images, synth_state_dict
m, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args("small"))
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0)); m = m.cuda().half().eval()
x = synth_images(32, 640, 640, seed=1).cuda().half()
out = m(x); pp = post["bbox"]; sizes = torch.tensor([[640.0, 640.0]] * 32, device="cuda")
for name, fn in (("PostProcess.select_packed B=32", lambda: pp.select_packed(out["pred_logits"], out["pred_boxes"], sizes)),):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); print(name, f"{ts[len(ts)//2]:.1f} us (min {ts[0]:.1f})")
PY
python tools/op_times.py --size small --batch 32 2>&1 | grep -i "topk\|rowmax\|Raw"
