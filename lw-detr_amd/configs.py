"""The five LW-DETR model configurations as ``build_model(args)`` namespaces.

Values restate the flags of the reference launch scripts (``scripts/lwdetr_{tiny,small,medium,
large,xlarge}_coco_eval.sh:9-31``) and the argparse defaults of ``main.py:39-183``; attribute names are
exactly the ones ``models/lwdetr.py:562-619``, ``models/backbone/__init__.py:38-63`` and
``models/transformer.py:541-564`` read, so the same namespace drives the reference and this package.
"""
from argparse import Namespace

_COMMON = dict(
    dataset_file="coco", device="cpu", position_embedding="sine", pretrained_encoder=None,
    drop_path=0.0, dropout=0.0, dim_feedforward=2048, dec_layers=3, group_detr=13, two_stage=True,
    bbox_reparam=True, lite_refpoint_refine=True, decoder_norm="LN", aux_loss=True,
    cls_loss_coef=2.0, bbox_loss_coef=5.0, giou_loss_coef=2.0, focal_alpha=0.25,
    set_cost_class=2.0, set_cost_bbox=5.0, set_cost_giou=2.0, sum_group_losses=False,
    use_varifocal_loss=False, use_position_supervised_loss=False, ia_bce_loss=False,
)

_SIZES = {
    "tiny": dict(encoder="vit_tiny", vit_encoder_num_layers=6, window_block_indexes=[0, 2, 4],
                 out_feature_indexes=[1, 3, 5], projector_scale=["P4"], hidden_dim=256, sa_nheads=8,
                 ca_nheads=16, dec_n_points=2, num_queries=100, num_select=100),
    "small": dict(encoder="vit_tiny", vit_encoder_num_layers=10, window_block_indexes=[0, 1, 3, 6, 7, 9],
                  out_feature_indexes=[2, 4, 5, 9], projector_scale=["P4"], hidden_dim=256, sa_nheads=8,
                  ca_nheads=16, dec_n_points=2, num_queries=300, num_select=300),
    "medium": dict(encoder="vit_small", vit_encoder_num_layers=10, window_block_indexes=[0, 1, 3, 6, 7, 9],
                   out_feature_indexes=[2, 4, 5, 9], projector_scale=["P4"], hidden_dim=256, sa_nheads=8,
                   ca_nheads=16, dec_n_points=2, num_queries=300, num_select=300),
    "large": dict(encoder="vit_small", vit_encoder_num_layers=10, window_block_indexes=[0, 1, 3, 6, 7, 9],
                  out_feature_indexes=[2, 4, 5, 9], projector_scale=["P3", "P5"], hidden_dim=384,
                  sa_nheads=12, ca_nheads=24, dec_n_points=4, num_queries=300, num_select=300),
    "xlarge": dict(encoder="vit_base", vit_encoder_num_layers=10, window_block_indexes=[0, 1, 3, 6, 7, 9],
                   out_feature_indexes=[2, 4, 5, 9], projector_scale=["P3", "P5"], hidden_dim=384,
                   sa_nheads=12, ca_nheads=24, dec_n_points=4, num_queries=300, num_select=300),
}

SIZES = tuple(_SIZES)

VIT_SIZES = {"vit_tiny": (192, 12), "vit_small": (384, 12), "vit_base": (768, 12)}   # models/backbone/backbone.py:46-51
LEVEL_SCALE = {"P3": 2.0, "P4": 1.0, "P5": 0.5}                                       # models/backbone/backbone.py:124-129

# Algorithmic GFLOP per image (2*MAC; GEMM + conv + attention bmm), SURVEY.md section 2.2 / BASELINE.md section 2.
GFLOP_PER_IMAGE = {("tiny", 640): 21.40, ("small", 640): 31.76, ("medium", 640): 83.93,
                   ("large", 640): 137.51, ("xlarge", 640): 342.51, ("xlarge", 960): 860.10}


def get_args(size: str, **overrides) -> Namespace:
    """Namespace for one of tiny/small/medium/large/xlarge (a fresh copy each call)."""
    if size not in _SIZES:
        raise KeyError(f"unknown LW-DETR size {size!r}; choose from {SIZES}")
    d = dict(_COMMON)
    d.update({k: (list(v) if isinstance(v, list) else v) for k, v in _SIZES[size].items()})
    d.update(overrides)
    return Namespace(**d)
