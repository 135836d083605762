"""GPU: parity of the HIP path at the BASELINE configurations AS STATED - model size x resolution x batch x dtype
(BASELINE.json configs 2-5; the per-GPU shard of the 8-GPU configs) - against the fp32 CPU oracle run on the same
synthetic batch in chunks on the host cores (tests/helpers.py:oracle_batch).

The model runs FREE (its own two-stage selection), then the oracle is run with that selection forced. Why this way
round: LW-DETR pairs the k-th selected token with the k-th learned query embedding (models/lwdetr.py:150-155,
models/transformer.py:246-264), so the ORDER of near-tied top-k scores changes every downstream tensor; 16-bit arithmetic
reorders near-ties (any 16-bit implementation does), which says nothing about the kernels. Checked per config:

  * selection: the oracle's own top-k set vs the model's (overlap), and - with the ORACLE's scores - how much worse the
    model's picks are than the oracle's (rank-wise score gap): a wrong token would show as a gap above 16-bit noise;
  * slot-wise: max / mean |difference| of final and encoder logits and boxes over all B x 300 slots;
  * detections: PostProcess on both sides (models/lwdetr.py:509-544); fraction of the oracle's 100 most confident
    detections per image that the model reports with the same label and every box coordinate within `px` pixels, and the
    score differences of those.

  * the yardstick (round 3): the reference arithmetic itself in the configuration's 16-bit dtype (the CPU oracle with every
    parameter and activation cast, helpers.oracle_lowp) on 8 images spread over both launch chains (round 5; rounds 3-4: the first
    two), same forced selection. Its distance
    to the fp32 oracle is what 16-bit arithmetic costs on this network; the HIP path's error on the SAME images must stay
    within 1.5x of it (it measures ~0.35x: f32 accumulators, LayerNorm statistics and softmax; fewer rounding points);
  * the auxiliary (per-decoder-layer) outputs are compared at the full batch as well.

The absolute bounds are <= 2x the value measured on MI355X (written next to them; each run writes
gpurun_out/parity_config_*.json, the tracked copy is profiles/parity_config_*.json)."""
import json
import os

import numpy as np
import pytest
import torch

import lwdetr_amd
from helpers import ROOT, box_iou_xyxy, oracle_batch, oracle_lowp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

#            name                 size      res  batch dtype            bounds (see _BOUNDS)
CONFIGS = [("tiny_b32_fp16", "tiny", 640, 32, torch.float16),            # a driver-visible `other_configs` line of bench.py since round 4
           ("small_b32_fp16", "small", 640, 32, torch.float16),
           ("medium_b64_bf16", "medium", 640, 64, torch.bfloat16),
           ("large_b32_fp16", "large", 640, 32, torch.float16),
           ("xlarge960_b16_fp16", "xlarge", 960, 16, torch.float16)]
# Measured on MI355X (round 2, profiles/parity_config_*.json); every bound is <= 2x its measurement:
#                         logit_max  box_max  logit_mean  overlap  gap     found   score
#   small  B32 fp16        0.0322    0.0035    0.00226    0.9978   0.0037  1.0     0.0030
#   medium B64 bf16        0.2224    0.0181    0.01797    0.9852   0.0421  0.9998  0.0192
#   large  B32 fp16        0.0416    0.0037    0.00264    0.9971   0.0049  1.0     0.0035
#   xlarge 960 B16 fp16    0.0569    0.0089    0.00322    0.9967   0.0026  1.0     0.0031
# (logits span about -9.7 .. -0.1 with std 1.0; boxes are cxcywh with the image = 1.) logit_max / box_max / logit_mean:
# slot-wise |difference| of final + encoder outputs; overlap / gap: two-stage selection vs the oracle's, judged with the
# oracle's scores; found: fraction of the oracle's detections reported with the same label and every box coordinate within
# `px` pixels of a 640 x 480 target; score: max |d score| over those.
_BOUNDS = {
    # tiny (round 5): measured on MI355X - see profiles/parity_config_tiny_b32_fp16.json; bounds <= 2x measured
    #   tiny   B32 fp16        0.0409    0.0053    0.00222    0.9978   0.0021  0.9975  0.0018
    "tiny_b32_fp16": dict(logit_max=0.08, box_max=0.0105, logit_mean=0.0044, overlap=0.99, gap=0.0042, found=0.98, score=0.0036, px=2.0),
    "small_b32_fp16": dict(logit_max=0.064, box_max=0.007, logit_mean=0.0045, overlap=0.99, gap=0.0075, found=0.98, score=0.006, px=2.0),
    "medium_b64_bf16": dict(logit_max=0.44, box_max=0.036, logit_mean=0.036, overlap=0.97, gap=0.084, found=0.98, score=0.038, px=8.0),
    "large_b32_fp16": dict(logit_max=0.083, box_max=0.0074, logit_mean=0.0052, overlap=0.99, gap=0.0098, found=0.98, score=0.007, px=2.0),
    # round 5: gap / score re-measured at 0.0042-0.0052 / 0.0040-0.0060 on three runs (with and without the opt-in LayerNorm fold; every
    # averaged and calibrated figure identical: logit_mean 0.00321-0.00323, ours / reference 16-bit 0.27): they are maxima over 4800 ranks /
    # 1600 detections and follow which near-ties reorder - bounds = 1.5x the largest measurement (profiles/r5d_*). Round 6: these two absolute
    # bounds are a regression fence only; the oracle-derived bound for both quantities is the calibrated one at the end of the test
    # (topk_score_gap / score_slot_max <= 1.5x the reference's own 16-bit arithmetic on the same images), for every configuration.
    "xlarge960_b16_fp16": dict(logit_max=0.11, box_max=0.0178, logit_mean=0.0064, overlap=0.99, gap=0.0078, found=0.98, score=0.009, px=2.0),
}


@pytest.mark.parametrize("name,size,res,batch,dtype", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_baseline_config_parity(name, size, res, batch, dtype):
    from lwdetr_amd.synth import synth_images, synth_state_dict
    cfg = lwdetr_amd.get_args(size)
    model, _, post = lwdetr_amd.build_model(cfg)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).to(dtype).eval()
    x = synth_images(batch, res, res, seed=4321).to(DEV).to(dtype)
    col = {}
    out = model(x, _collect=col)                                            # free-running, in the launch plan bench.py times:
    expect_chains = type(model)._chains_for(batch, res, res)                          # two chains from 32 images (each chain's selection
    assert col["launch_chains"] == expect_chains, col["launch_chains"]      # is collected and concatenated)
    ours = col["topk_idx"].cpu().numpy()
    sizes = torch.tensor([[480.0, 640.0]] * batch, device=DEV)
    res_ = post["bbox"](out, sizes)
    exp = oracle_batch(size, batch, res, img_seed=4321, forced_topk=ours)   # fp32 CPU oracle on the model's selection
    # ---- selection, judged with the oracle's scores
    ref_sc = exp["enc_class_max"]
    ref_idx = np.argsort(-ref_sc, axis=1, kind="stable")[:, :cfg.num_queries]
    ov = np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(ours, ref_idx)])
    gap = np.abs(np.sort(np.take_along_axis(ref_sc, ours, 1), 1) - np.sort(np.take_along_axis(ref_sc, ref_idx, 1), 1)).max()
    # ---- slot-wise
    dl = np.abs(out["pred_logits"].float().cpu().numpy() - exp["pred_logits"])
    db = np.abs(out["pred_boxes"].float().cpu().numpy() - exp["pred_boxes"])
    del_ = np.abs(out["enc_outputs"]["pred_logits"].float().cpu().numpy() - exp["enc_logits"])
    deb = np.abs(out["enc_outputs"]["pred_boxes"].float().cpu().numpy() - exp["enc_boxes"])
    m = {"logit_max": float(max(dl.max(), del_.max())), "box_max": float(max(db.max(), deb.max())),
         "logit_mean": float(dl.mean()), "box_mean": float(db.mean()),
         "logit_range": [float(exp["pred_logits"].min()), float(exp["pred_logits"].max())],
         "logit_std": float(exp["pred_logits"].std()), "topk_set_overlap": float(ov), "topk_score_gap": float(gap)}
    # a detection of the oracle counts as FOUND when the model reports the same label with every box coordinate within
    # `px` pixels (pixel distance, not IoU: with random weights many boxes are a few pixels wide and IoU is hypersensitive)
    found, dscore, total, ious, near_ds = 0, 0.0, 0, [], []
    px = _BOUNDS[name]["px"]
    for i in range(batch):
        s_o, l_o, b_o = exp["post_scores"][i], exp["post_labels"][i], exp["post_boxes"][i]
        s_m = res_[i]["scores"].float().cpu().numpy()
        l_m = res_[i]["labels"].cpu().numpy()
        b_m = res_[i]["boxes"].float().cpu().numpy()
        top = np.argsort(-s_o)[:100]                           # the oracle's 100 most confident detections of the image
        dist = np.abs(b_o[top][:, None, :] - b_m[None, :, :]).max(-1)
        dist = np.where(l_o[top][:, None] == l_m[None, :], dist, np.inf)
        # several detections of one label can sit within `px` of each other (the same object seen by neighbouring queries): among
        # the candidates inside the box tolerance the partner is the one with the closest score - taking the nearest box made
        # the score column jump (0.018 -> 0.082 on medium bf16) on a 0.003 change of one logit while every slot-wise figure stood
        cand = dist <= px
        ok = cand.any(1)
        ds = np.where(cand, np.abs(s_o[top][:, None] - s_m[None, :]), np.inf)
        j = ds.argmin(1)
        total += len(top)
        found += int(ok.sum())
        if ok.any():
            dscore = max(dscore, float(ds[np.arange(len(top)), j][ok].max()))
            jn = dist.argmin(1)                                   # the other pairing: nearest box - reported and bounded too
            near_ds.append(np.abs(s_o[top] - s_m[jn])[ok])
            iou = box_iou_xyxy(b_o[top][ok], b_m[j][ok])
            ious.append(np.diag(iou))
    ious = np.concatenate(ious) if ious else np.zeros(1)
    near_ds = np.concatenate(near_ds) if near_ds else np.zeros(1)
    m.update({"found": found / total, "score": dscore, "match_px": px, "launch_chains": col["launch_chains"],
              "score_nearest_box_max": float(near_ds.max()), "score_nearest_box_p999": float(np.percentile(near_ds, 99.9)),
              "iou_of_found_median": float(np.median(ious)), "iou_of_found_p10": float(np.percentile(ious, 10)),
              "config": {"size": size, "res": res, "batch": batch, "dtype": str(dtype).split(".")[-1]}})
    # ---- auxiliary outputs (decoder layers 0 .. L-2) at the full batch
    if out.get("aux_outputs"):
        al = np.stack([a["pred_logits"].float().cpu().numpy() for a in out["aux_outputs"]], 1)
        ab = np.stack([a["pred_boxes"].float().cpu().numpy() for a in out["aux_outputs"]], 1)
        m["aux_logit_max"] = float(np.abs(al - exp["aux_logits"]).max())
        m["aux_box_max"] = float(np.abs(ab - exp["aux_boxes"]).max())
    # ---- the yardstick: the reference arithmetic in this dtype (CPU) on 8 images spread over BOTH launch chains (VERDICT r4 item 6a:
    # the round-3/4 sample was the first two images = chain 0 only), same forced selection
    half = batch // 2
    lo_idx = sorted({0, 1, half // 2, half - 1, half, half + 1, half + half // 2, batch - 1})
    low = oracle_lowp(size, len(lo_idx), res, 4321, dtype, ours, idx=lo_idx)
    sel = np.asarray(lo_idx)
    ref16 = {"logit_max": float(max(np.abs(low["pred_logits"] - exp["pred_logits"][sel]).max(), np.abs(low["enc_logits"] - exp["enc_logits"][sel]).max())),
             "logit_mean": float(np.abs(low["pred_logits"] - exp["pred_logits"][sel]).mean()),
             "box_max": float(max(np.abs(low["pred_boxes"] - exp["pred_boxes"][sel]).max(), np.abs(low["enc_boxes"] - exp["enc_boxes"][sel]).max())),
             "box_mean": float(np.abs(low["pred_boxes"] - exp["pred_boxes"][sel]).mean())}
    ours16 = {"logit_max": float(max(dl[sel].max(), del_[sel].max())), "logit_mean": float(dl[sel].mean()),
              "box_max": float(max(db[sel].max(), deb[sel].max())), "box_mean": float(db[sel].mean())}
    # Round 6: the two figures that had only absolute bounds - the rank-wise score gap of the two-stage selection and the
    # detection score error - calibrated the same way. Selection: the 16-bit reference arithmetic's OWN free selection (top-k of its class maxima)
    # judged with the fp32 oracle's scores, against ours on the same images. Scores: sigmoid of the final logits slot by slot (the selection is
    # forced, so slots correspond) - what PostProcess turns into detection scores, without the matching step's twin-detection noise.
    nq_ = cfg.num_queries
    low_idx = np.argsort(-low["enc_class_max"], axis=1, kind="stable")[:, :nq_]
    srt = lambda pick: np.sort(np.take_along_axis(ref_sc[sel], pick, 1), 1)
    ref16["topk_score_gap"] = float(np.abs(srt(low_idx) - srt(ref_idx[sel])).max())
    ours16["topk_score_gap"] = float(np.abs(srt(ours[sel]) - srt(ref_idx[sel])).max())
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    ref16["score_slot_max"] = float(np.abs(sig(low["pred_logits"]) - sig(exp["pred_logits"][sel])).max())
    ours16["score_slot_max"] = float(np.abs(sig(out["pred_logits"].float().cpu().numpy()[sel]) - sig(exp["pred_logits"][sel])).max())
    # per launch chain: the same ratio on the images of each half batch (a chain-1-only defect must not hide in the pooled figure)
    per_chain = {}
    for ci, part in enumerate((sel[sel < half], sel[sel >= half])):
        pos = np.searchsorted(sel, part)
        r_ = max(np.abs(low["pred_logits"][pos] - exp["pred_logits"][part]).max(), np.abs(low["enc_logits"][pos] - exp["enc_logits"][part]).max())
        o_ = max(dl[part].max(), del_[part].max())
        per_chain[f"chain{ci}"] = {"images": [int(i) for i in part], "ref_logit_max": float(r_), "ours_logit_max": float(o_), "ratio": round(float(o_ / max(r_, 1e-12)), 3)}
    m["lowp_images"] = [int(i) for i in lo_idx]
    m["ours_over_ref_16bit_per_chain"] = per_chain
    m["ref_16bit_err"] = ref16
    m["ours_same_images"] = ours16
    m["ours_over_ref_16bit"] = {k: round(ours16[k] / max(ref16[k], 1e-12), 3) for k in ref16}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"parity_config_{name}.json"), "w") as f:
        json.dump(m, f, indent=1)
    b = _BOUNDS[name]
    assert torch.isfinite(out["pred_logits"].float()).all()
    assert m["topk_set_overlap"] > b["overlap"] and m["topk_score_gap"] < b["gap"], m
    assert m["logit_max"] < b["logit_max"] and m["box_max"] < b["box_max"] and m["logit_mean"] < b["logit_mean"], m
    assert m["found"] > b["found"] and m["score"] < b["score"], m
    # nearest-box pairing: all but one detection in a thousand within the same score bound (a twin detection of the same label a
    # pixel away may pair with its neighbour's score - that is what the closest-score pairing above removes, not a kernel error)
    assert m["score_nearest_box_p999"] < b["score"], m
    # calibrated: no worse than 1.5x what the reference's own arithmetic costs in this dtype, on the same images
    for k in ("logit_max", "logit_mean", "box_max", "box_mean", "score_slot_max"):
        assert ours16[k] <= 1.5 * ref16[k], (k, ours16, ref16)
    # the selection gap is a maximum over a few thousand ranks of which near-ties reorder: 1.5x the reference arithmetic's own, with a floor of one
    # 16-bit ulp of a class logit around -2 (a gap below that is a tie in the dtype)
    assert ours16["topk_score_gap"] <= max(1.5 * ref16["topk_score_gap"], 2e-3), (ours16, ref16)
    for ck, cv in per_chain.items():
        assert cv["ours_logit_max"] <= 1.5 * cv["ref_logit_max"], (ck, cv)
    if "aux_logit_max" in m:
        assert m["aux_logit_max"] < 1.5 * b["logit_max"] and m["aux_box_max"] < 1.5 * b["box_max"], m
