"""``LWDETR`` / ``PostProcess`` / ``build`` with the reference's API (``models/lwdetr.py:36-216, 509-544, 562-619``).

The module tree (and therefore the state dict) is the reference's; ``forward`` runs the HIP launch plan of
``lwdetr_amd.engine``. There is no eager / CPU implementation of the forward in this package: on a non-ROCm device,
or without ``liblwdetr_hip.so``, ``forward`` raises.
"""
import copy
import os

import torch
from torch import nn

from .. import _native
from ..engine import ForwardPlan, PackedWeights
from ..plan_cache import PlanCache
from .modules import (MLP, Backbone, Joiner, PositionEmbeddingSine, Transformer, focal_prior_bias)
from .nested import NestedTensor, nested_tensor_from_tensor_list

# Dense batches of >= _TWO_STREAM_MIN_BATCH images run as two launch chains on two streams (LWDETR._forward_chains):
# LWDETR_STREAMS=1 turns that off, =n runs n chains whenever the parts have >= 8 images; set_streams() is the run-time switch.
_STREAMS = int(os.environ.get("LWDETR_STREAMS", "0"))
_TWO_STREAM_MIN_BATCH = 32


def set_streams(n):
    """0 = default policy (two chains from _TWO_STREAM_MIN_BATCH images), 1 = always one chain, n >= 2 = n chains (parts of >= 8 images)."""
    global _STREAMS
    _STREAMS = int(n)



class LWDETR(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, group_detr=1,
                 two_stage=False, lite_refpoint_refine=False, bbox_reparam=False, args=None):
        super().__init__()
        if not (two_stage and lite_refpoint_refine and bbox_reparam):
            raise NotImplementedError("lwdetr_amd implements the published LW-DETR configuration: --two_stage "
                                      "--bbox_reparam --lite_refpoint_refine (all five model sizes use it)")
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = nn.Linear(hidden_dim, num_classes)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.refpoint_embed = nn.Embedding(num_queries * group_detr, 4)
        self.query_feat = nn.Embedding(num_queries * group_detr, hidden_dim)
        nn.init.constant_(self.refpoint_embed.weight.data, 0)
        self.backbone = backbone
        self.aux_loss, self.group_detr = aux_loss, group_detr
        self.lite_refpoint_refine, self.bbox_reparam, self.two_stage = lite_refpoint_refine, bbox_reparam, two_stage
        self.transformer.decoder.bbox_embed = None
        self.class_embed.bias.data = focal_prior_bias(num_classes)
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        self.transformer.enc_out_bbox_embed = nn.ModuleList(copy.deepcopy(self.bbox_embed) for _ in range(group_detr))
        self.transformer.enc_out_class_embed = nn.ModuleList(copy.deepcopy(self.class_embed) for _ in range(group_detr))
        self._export = False
        self._args = copy.copy(args)
        self._packed = None      # PackedWeights for the current (device, dtype, parameter versions)
        self._plans = PlanCache()         # (B, H, W, slot) -> ForwardPlan, bounded (plan_cache.py)
        self._tok_tensors = None
        self._side_streams = {}  # device -> side streams of the launch chains (a stream belongs to one device)

    # ---- cache invalidation: any change of device / dtype / parameter values drops the packed weights
    def invalidate_cache(self):
        self._packed, self._plans, self._tok_tensors = None, PlanCache(), None
        self._side_streams = {}

    def _apply(self, fn, *a, **k):
        self.invalidate_cache()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_cache()
        return super().load_state_dict(*a, **k)

    def _weights_token(self):
        """Identity of everything the packed weights were derived from: storage, dtype, device and in-place version of
        every parameter AND buffer (BatchNorm running statistics are folded into the conv weights). Catches
        ``model.backbone.half()`` / ``.to()`` on a sub-module (parameter storage moves), ``p.data = ...`` and in-place
        updates alike. (Re-binding a buffer attribute to a new tensor needs an explicit ``invalidate_cache()``.)"""
        if self._tok_tensors is None:       # flat list, rebuilt after _apply / load_state_dict on this module (~0.08 ms / call)
            self._tok_tensors = list(self.parameters()) + list(self.buffers())
        p = self.class_embed.weight
        return (p.device, p.dtype, hash(tuple([(t.data_ptr(), t._version) for t in self._tok_tensors])))

    def _packed_weights(self):
        if self.training:
            raise NotImplementedError("lwdetr_amd implements the inference forward; call model.eval()")
        tok = self._weights_token()
        if self._packed is None or self._packed[0] != tok:
            dev, dt = tok[0], tok[1]
            if dev.type != "cuda":
                raise _native.NativeError("lwdetr_amd: the forward path runs on a ROCm device only - move the model "
                                          "with .to('cuda'); there is no CPU implementation")
            mixed = {t.dtype for t in self.parameters() if t.is_floating_point()} | {t.device for t in self.parameters()}
            if len(mixed) != 2:
                raise _native.NativeError(f"lwdetr_amd: all parameters must share one dtype and device (found {mixed})")
            _native.lib()       # raises loudly when the extension is missing
            self._packed = (tok, PackedWeights(dict(self.state_dict()), self._args, dev, dt))
            self._plans = PlanCache()
        return self._packed

    def _plan(self, b, h, w, private=False, slot=0):
        """Launch plan for a batch shape. ``private`` plans are not cached: a HIP graph owns its plan's buffers and padding
        state, which eager calls of the same shape must never touch."""
        tok, pw = self._packed_weights()
        key = (b, h, w, slot)
        if private:
            with torch.cuda.device(tok[0]):
                return ForwardPlan(pw, b, h, w)
        def build():
            with torch.cuda.device(tok[0]):
                return ForwardPlan(pw, b, h, w)
        # bounded by the TOTAL number of resident plans (default 8, LWDETR_PLAN_CACHE): least recently used shape first, all launch
        # chains of a shape together (lwdetr_amd/plan_cache.py; INTEGRATION.md section 6 for the memory this holds)
        return self._plans.get(key, build)

    @torch.no_grad()
    def forward(self, samples, targets=None, _forced_topk=None, _collect=None):
        """samples: NestedTensor | list[Tensor(3,h,w)] | Tensor(B,3,H,W). Returns the reference's output dict:
        pred_logits (B,nq,C), pred_boxes (B,nq,4) cxcywh, aux_outputs (dec_layers-1 dicts), enc_outputs."""
        if isinstance(samples, torch.Tensor):
            x, mask = samples, None                     # a dense batch has no padding: no mask work, no host sync
        else:
            if isinstance(samples, list):
                same = all(t.shape == samples[0].shape for t in samples)
                samples = nested_tensor_from_tensor_list(samples)
                if same:
                    samples.mask = None
            x, mask = samples.tensors, samples.mask
        b, _, h, w = x.shape
        nch = self._chains_for(b, h, w) if (mask is None and _forced_topk is None and isinstance(samples, torch.Tensor)) else 1
        if nch > 1:
            return self._forward_chains(x, b, h, w, nch, collect=_collect)
        plan = self._plan(b, h, w)
        with torch.cuda.device(plan.dev):
            return plan.run(x, mask, forced_topk=_forced_topk, collect=_collect)

    @torch.no_grad()
    def detect(self, images, target_sizes, postprocess):
        """``forward`` + ``postprocess.select_packed`` as one call on a dense batch: returns (output dict, (B, K, 6) f32 records).
        Same launches and values as the two calls; with launch chains every chain selects its own images on its own stream
        (the selection is one workgroup per image and would otherwise run alone at the end of the step)."""
        assert isinstance(images, torch.Tensor) and images.dim() == 4
        b, _, h, w = images.shape
        nch = self._chains_for(b, h, w)
        if nch > 1:
            return self._forward_chains(images, b, h, w, nch, post=(postprocess, target_sizes))
        out = self.forward(images)
        return out, postprocess.select_packed(out["pred_logits"], out["pred_boxes"], target_sizes)

    @staticmethod
    def _chains_for(b, h=640, w=640):
        """Launch chains for a dense batch of b images (h x w pixels: kept in the signature, not used by the default rule - xlarge 960 x 960 B = 16
        as two 8-image chains measured +1.1 %, +1.3 % and -1.0 % on three boxes of round 5: inside the noise, left at one chain as in round 4):
        default two from _TWO_STREAM_MIN_BATCH images; LWDETR_STREAMS / set_streams: 1 = one chain, n >= 2 = n chains whenever the batch splits
        into n parts of at least 8 images."""
        if _STREAMS == 1:
            return 1
        if _STREAMS >= 2:
            return _STREAMS if (b % _STREAMS == 0 and b // _STREAMS >= 8) else 1
        return 2 if (b >= _TWO_STREAM_MIN_BATCH and b % 2 == 0) else 1

    def _forward_chains(self, x, b, h, w, nch, post=None, collect=None):
        """The parts of a dense batch as ``nch`` launch chains on ``nch`` streams. Every kernel of the path runs its workgroups in
        lockstep (all of them load, then all compute, then all store - DESIGN.md section 5b); with other chains a few kernels
        ahead or behind, one part's bandwidth-bound phases and vector-bound attention run beside another part's matrix phases
        (measured with two chains: medium B = 64 bf16 6.70 k -> 7.34 k img/s, large B = 32 fp16 3.59 k -> 4.06 k, small B = 32
        fp16 11.66 k -> 11.78 k). Images are independent: each part is computed exactly as a batch of b / nch images is, and
        writes its rows of the call's output tensors (allocated on the current stream, which waits for every chain)."""
        part = b // nch
        plans = [self._plan(part, h, w, slot=i) for i in range(nch)]
        dev = plans[0].dev
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            sides = self._side_streams.setdefault(dev, [])
            while len(sides) < nch - 1:
                sides.append(torch.cuda.Stream(dev))
            assert all(s_.device == dev for s_ in sides[:nch - 1])
            outs = plans[0].alloc_outputs(b)        # every chain writes its images' rows of ONE set of output tensors
            det = None
            if post is not None:
                pp, sizes = post
                sizes = sizes.to(device=dev, dtype=torch.float32).contiguous()
                det = torch.empty(b, pp.num_select, 6, dtype=torch.float32, device=dev)

            def chain(i):
                plans[i].run(x[i * part:(i + 1) * part], None, into=(outs, i * part))
                if post is not None:
                    sl = slice(i * part, (i + 1) * part)
                    pp.select_packed(outs[2][-1, sl], outs[3][-1, sl], sizes[sl], out=det[sl])

            for i in range(1, nch):
                side = sides[i - 1]
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    chain(i)
            chain(0)
            for i in range(1, nch):
                cur.wait_stream(sides[i - 1])
            if collect is not None:      # tests: the chains' intermediate buffers, concatenated to what one full-batch plan holds
                S, d = plans[0].S, plans[0].d
                cat = lambda f, dim=0: torch.cat([f(p) for p in plans], dim)
                collect.update({"topk_idx": cat(lambda p: p.topk_idx), "enc.class_max": cat(lambda p: p.cls_max),
                                "memory": cat(lambda p: p.memory.view(part, S, d)), "taps_cat": cat(lambda p: p.taps_cat),
                                "x": cat(lambda p: p.x), "hs": cat(lambda p: p.hs, 1), "om": cat(lambda p: p.om.view(part, S, d)),
                                "launch_chains": nch})
            res = plans[0].output_dict(*outs)
            return res if post is None else (res, det)

    @torch.no_grad()
    def capture(self, images, postprocess=None, target_sizes=None):
        """HIP-graph the forward for one dense batch shape (the launch-bound bs=1 latency path; the reference reaches its
        latency numbers through a TensorRT engine, ``deploy/benchmark.py``). ``images`` (B,3,H,W) fixes shape, device and
        dtype. Returns ``GraphedForward``: ``g(images)`` copies the batch into the static input, replays the captured
        launches and returns the output dict (static tensors, overwritten by the next replay); with ``postprocess``
        (a ``PostProcess``) and ``target_sizes`` the detections are captured too and returned as
        ``(scores, labels, boxes)``."""
        from ..engine import GraphedForward
        assert isinstance(images, torch.Tensor) and images.dim() == 4
        b, _, h, w = images.shape
        return GraphedForward(self._plan(b, h, w, private=True), images, postprocess, target_sizes)

    def export(self):
        """Export-mode forward of the reference (``lwdetr.py:103-109, 176-195``): tensor in, (coords, logits) out."""
        self._export = True
        self._forward_origin = self.forward
        self.forward = self.forward_export

    @torch.no_grad()
    def forward_export(self, tensors):
        out = self._forward_origin(tensors)
        return out["pred_boxes"], out["pred_logits"]

    def update_drop_path(self, drop_path_rate, vit_encoder_num_layers):
        pass    # training-only knob of the reference (lwdetr.py:205-210); DropPath is identity at inference

    def update_dropout(self, drop_rate):
        for module in self.transformer.modules():
            if isinstance(module, nn.Dropout):
                module.p = drop_rate


class PostProcess(nn.Module):
    """Model output -> per-image {scores, labels, boxes(xyxy, absolute)} (reference ``lwdetr.py:509-544``)."""

    def __init__(self, num_select=300):
        super().__init__()
        self.num_select = num_select

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(logits) == len(target_sizes) and target_sizes.shape[1] == 2
        scores, labels, xyxy = self.select(logits, boxes, target_sizes)
        scores = scores.to(logits.dtype)                # as the reference: sigmoid of the model-dtype logits
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, xyxy)]

    def select(self, logits, boxes, target_sizes):
        """Batched form: (scores (B,K), labels (B,K) int64, boxes (B,K,4) xyxy pixels). Tensors on the GPU go through the
        fused HIP kernel (``lwdetr_postprocess``: radix top-k on the logits, sigmoid / box conversion of the K winners,
        scores and boxes in f32); host tensors (the reference's CPU evaluation plumbing) use the tensor ops below."""
        if logits.is_cuda:
            return self._select_hip(logits, boxes, target_sizes)
        prob = logits.sigmoid()
        scores, idx = torch.topk(prob.view(logits.shape[0], -1), self.num_select, dim=1)
        box_idx = idx // logits.shape[2]
        labels = idx % logits.shape[2]
        cx, cy, w, h = boxes.unbind(-1)
        w, h = w.clamp(min=0.0), h.clamp(min=0.0)       # util/box_ops.py:21-25
        xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
        xyxy = torch.gather(xyxy, 1, box_idx.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        xyxy = xyxy * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        return scores, labels, xyxy

    def select_packed(self, logits, boxes, target_sizes, out=None):
        """``select`` written as ONE (B, K, 6) f32 tensor of rows (score, label, x0, y0, x1, y1) - the record
        ``lwdetr_amd.dist.all_gather_detections`` ships (same kernel, same values as ``dist.pack_detections(*select(...))``).
        ``out``: a contiguous (B, K, 6) f32 tensor (or a leading-dimension slice of one) to write into."""
        if not logits.is_cuda:
            from ..dist import pack_detections
            res = pack_detections(*self.select(logits, boxes, target_sizes))
            return res if out is None else out.copy_(res)
        from .. import _native
        b, nq, ncls = logits.shape
        dev = logits.device
        logits = logits.contiguous()
        boxes = boxes.to(logits.dtype).contiguous()
        sizes = target_sizes.to(device=dev, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty(b, self.num_select, 6, dtype=torch.float32, device=dev)
        assert out.shape == (b, self.num_select, 6) and out.dtype == torch.float32 and out.is_contiguous() and out.device == dev
        with torch.cuda.device(dev):
            rc = _native.lib().lwdetr_postprocess_packed(logits.data_ptr(), boxes.data_ptr(), sizes.data_ptr(), b, nq, ncls,
                                                         self.num_select, out.data_ptr(), _native.dtype_code(logits.dtype),
                                                         _native.stream_ptr(dev))
        _native.check(rc, "lwdetr_postprocess_packed")
        return out

    def _select_hip(self, logits, boxes, target_sizes, out=None):
        from .. import _native
        b, nq, ncls = logits.shape
        k, dev = self.num_select, logits.device
        logits = logits.contiguous()
        boxes = boxes.to(logits.dtype).contiguous()
        sizes = target_sizes.to(device=dev, dtype=torch.float32).contiguous()
        if out is None:
            out = (torch.empty(b, k, dtype=torch.float32, device=dev), torch.empty(b, k, dtype=torch.int64, device=dev),
                   torch.empty(b, k, 4, dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            rc = _native.lib().lwdetr_postprocess(logits.data_ptr(), boxes.data_ptr(), sizes.data_ptr(), b, nq, ncls, k,
                                                  out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                  _native.dtype_code(logits.dtype), _native.stream_ptr(dev))
        _native.check(rc, "lwdetr_postprocess")
        return out


def build(args):
    """``build_model(args) -> (model, criterion, postprocessors)`` (reference ``lwdetr.py:562-619``).

    The training criterion (Hungarian matcher + losses, ``lwdetr.py:218-506``, ``matcher.py``) is outside this
    package's scope: when the reference's ``models.lwdetr.SetCriterion`` is importable (i.e. this package is dropped
    into the reference repository) it is built exactly as the reference does; otherwise ``criterion`` is ``None``."""
    num_classes = 20 if args.dataset_file != "coco" else 91
    if args.dataset_file == "o365":
        num_classes = 366
    if getattr(args, "position_embedding", "sine") != "sine":
        raise NotImplementedError("only the sine position embedding of the published configs is supported")
    backbone = Joiner(
        Backbone(args.encoder, args.vit_encoder_num_layers, args.window_block_indexes, args.hidden_dim,
                 args.out_feature_indexes, args.projector_scale),
        PositionEmbeddingSine(args.hidden_dim // 2))
    args.num_feature_levels = len(args.projector_scale)
    transformer = Transformer(
        d_model=args.hidden_dim, sa_nhead=args.sa_nheads, ca_nhead=args.ca_nheads, num_queries=args.num_queries,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        group_detr=args.group_detr, two_stage=getattr(args, "two_stage", False),
        num_feature_levels=args.num_feature_levels, dec_n_points=args.dec_n_points)
    if getattr(args, "decoder_norm", "LN") != "LN":
        raise NotImplementedError("decoder_norm must be 'LN' (the published configs)")
    model = LWDETR(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries,
                   aux_loss=args.aux_loss, group_detr=args.group_detr, two_stage=args.two_stage,
                   lite_refpoint_refine=args.lite_refpoint_refine, bbox_reparam=args.bbox_reparam, args=args)
    model.eval()
    criterion = _reference_criterion(args, num_classes)
    postprocessors = {"bbox": PostProcess(num_select=args.num_select)}
    return model, criterion, postprocessors


def _reference_criterion(args, num_classes):
    try:
        from models.lwdetr import SetCriterion      # the reference's own training criterion, if co-installed
        from models.matcher import build_matcher
    except Exception:
        return None
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef, "loss_giou": args.giou_loss_coef}
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        if args.two_stage:
            aux.update({k + "_enc": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    crit = SetCriterion(num_classes, matcher=build_matcher(args), weight_dict=weight_dict,
                        focal_alpha=args.focal_alpha, losses=["labels", "boxes", "cardinality"],
                        group_detr=args.group_detr, sum_group_losses=getattr(args, "sum_group_losses", False),
                        use_varifocal_loss=args.use_varifocal_loss,
                        use_position_supervised_loss=args.use_position_supervised_loss, ia_bce_loss=args.ia_bce_loss)
    crit.to(torch.device(args.device))
    return crit
