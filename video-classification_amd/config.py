"""Config surface of the reference, drop-in: same keys, same defaults, same merge order.

Mirrors /root/reference/config/defaults.py:4-60 (yacs CfgNode tree, get_cfg, get_override_cfg) and
/root/reference/config/crop_cfg.py:22-57 (crop folder -> pixel size).  yacs is not installed in this image, so
``CfgNode`` is a small compatible subset: attribute access, clone(), merge_from_file(yaml), merge_from_list().
New keys (MODEL.DTYPE, MODEL.ARCH, MODEL.DEPTH, DIST.*) default to the reference's behaviour (fp32, its own geometry, 1 process).
"""
from __future__ import annotations

import copy
from pathlib import Path

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{path}{k} must be a mapping")
                self[k]._merge(v, f"{path}{k}.")
            else:
                old = self[k]
                if isinstance(old, float) and isinstance(v, str):
                    v = float(v)          # PyYAML reads '2e-4' as a string; yacs casts it back
                if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                    v = float(v)
                if old is not None and v is not None and type(old) is not type(v):
                    raise ValueError(f"Type mismatch for {path}{k}: {type(old).__name__} vs {type(v).__name__}")
                self[k] = v

    def merge_from_file(self, path) -> None:
        with open(str(path), "r") as f:
            data = yaml.safe_load(f) or {}
        self._merge(data)

    def merge_from_list(self, kv) -> None:
        assert len(kv) % 2 == 0
        for k, v in zip(kv[0::2], kv[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node._merge({parts[-1]: v}, ".".join(parts[:-1]) + ("." if len(parts) > 1 else ""))


_C = CfgNode()
_C.CHALEARN = CfgNode()
_C.DEBUG = False
_C.CHALEARN.ROOT = '/media/zc/C2000Pro-1TB/ChaLearnIsoAllClass'
_C.CHALEARN.NUM_CLASS = 249
_C.CHALEARN.BATCH_SIZE = 10
_C.CHALEARN.ISO = '0_Iso'
_C.CHALEARN.SAMPLE = '1_Sample'
_C.CHALEARN.SAMPLE_CLASS = 249
_C.CHALEARN.IMG = '2_Images'
_C.CHALEARN.IMG_SAMPLE_INTERVAL = 5
_C.CHALEARN.PAD = '3_Pad'
_C.CHALEARN.IUV = '4_IUV'
_C.CHALEARN.CSE = '4_CSE'
_C.CHALEARN.CROP_BODY = 'CropBody'
_C.CHALEARN.CLIP_LEN = 20
_C.CHALEARN.FLOW = '2_Flow'
_C.CHALEARN.FLOW_NPY = '2_Flow_npy'
_C.CHALEARN.IMG_ENERGY = '2_Images_energy'
_C.CHALEARN.FLOW_VIDEO = '2_Flow_New'
_C.CHALEARN.IUV_NEW = '4_IUV_New'
_C.CHALEARN.UV_VIDEO = '5_UV_Video'
_C.CHALEARN.BOX = '6_Box'
_C.DENSEPOSE = './detectron2/projects/DensePose'
_C.MODEL = CfgNode()
_C.MODEL.LOGS = 'logs'
_C.MODEL.NAME = 'new_feature_test'
_C.MODEL.CKPT_DIR = 'checkpoints'
_C.MODEL.R3D_INPUT = 'CropHTAH'
_C.MODEL.LR = 5e-4
_C.MODEL.FUSE = True
_C.MODEL.MAX_EPOCH = 100
_C.MODEL.INPUT_SIZE = 192
_C.NUM_CPU = 18
# ---- keys added by this engine; defaults reproduce the reference
_C.MODEL.DTYPE = 'fp32'        # 'fp32' (reference precision) | 'bf16' (benchmark precision)
_C.MODEL.ARCH = 'ref'          # 'ref' = init_my_slowfast geometry | 'canonical8x8' = SlowFast-R50 8x8 (BGR frames, PackPathway alpha 4)
_C.MODEL.DEPTH = 50            # the reference hard-codes 50 (model/my_slowfast.py:98); 18 / 26 = the small test networks
_C.DIST = CfgNode()
_C.DIST.BUCKET_MB = 32         # gradient all-reduce bucket size


def get_cfg() -> CfgNode:
    """A copy of the defaults (reference config/defaults.py:50-54)."""
    return _C.clone()


def get_override_cfg() -> CfgNode:
    """Defaults + ../cfg_override.yaml when present (reference config/defaults.py:56-61)."""
    cfg = get_cfg()
    override = Path('..', 'cfg_override.yaml')
    if override.is_file():
        cfg.merge_from_file(override)
    return cfg


# crop folder -> square size in pixels (reference config/crop_cfg.py:18-55)
crop_resize_dict = {'CropHTAH': 192, 'CropLHand': 64, 'CropRHand': 64, 'CropLHandArm': 128, 'CropRHandArm': 128,
                    'CropTorso': 128}
crop_folder_list = list(crop_resize_dict.keys())
