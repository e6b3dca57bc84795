"""Default values for the config keys the JTSM path reads, as one nested literal.  Values are those of
detectron2/config/defaults.py (e.g. FREEZE_AT :130, RESNETS.NORM :471, STRIDE_IN_1X1 :479,
DEFORM_ON_PER_STAGE :491) and, for the WSL block, projects/WSL/wsl/config/defaults.py:7-73."""
from .config import CfgNode as CN

_DEFAULTS = {
    "VERSION": 2,
    "OUTPUT_DIR": "./output",
    "VIS_PERIOD": 0,
    "SEED": -1,
    # detectron2/config/defaults.py:40-60 (INPUT), :84-96 (DATASETS), :607-620 (TEST.AUG)
    "INPUT": {"FORMAT": "BGR", "MASK_FORMAT": "polygon", "MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice",
              "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333},
    "DATASETS": {"PRECOMPUTED_PROPOSAL_TOPK_TRAIN": 2000, "PRECOMPUTED_PROPOSAL_TOPK_TEST": 1000},
    "TEST": {"DETECTIONS_PER_IMAGE": 100,
             "AUG": {"ENABLED": False, "MIN_SIZES": (400, 500, 600, 700, 800, 900, 1000, 1100, 1200), "MAX_SIZE": 4000,
                     "FLIP": True}},
    "SOLVER": {"MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0001,
               "WEIGHT_DECAY_NORM": 0.0, "BIAS_LR_FACTOR": 1.0, "WEIGHT_DECAY_BIAS": 0.0001, "IMS_PER_BATCH": 16},
    "MODEL": {
        "LOAD_PROPOSALS": False, "MASK_ON": False, "KEYPOINT_ON": False, "DEVICE": "cuda",
        "META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHTS": "",
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
        "FPN": {"IN_FEATURES": [], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
        "PROPOSAL_GENERATOR": {"NAME": "RPN", "MIN_SIZE": 0},
        # detectron2/config/defaults.py:156-222 (anchor generator, RPN)
        "ANCHOR_GENERATOR": {"NAME": "DefaultAnchorGenerator", "SIZES": [[32, 64, 128, 256, 512]],
                             "ASPECT_RATIOS": [[0.5, 1.0, 2.0]], "ANGLES": [[-90, 0, 90]], "OFFSET": 0.0},
        "RPN": {"HEAD_NAME": "StandardRPNHead", "IN_FEATURES": ["res4"], "BOUNDARY_THRESH": -1,
                "IOU_THRESHOLDS": [0.3, 0.7], "IOU_LABELS": [0, -1, 1], "BATCH_SIZE_PER_IMAGE": 256,
                "POSITIVE_FRACTION": 0.5, "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0), "SMOOTH_L1_BETA": 0.0, "LOSS_WEIGHT": 1.0,
                "PRE_NMS_TOPK_TRAIN": 12000, "PRE_NMS_TOPK_TEST": 6000, "POST_NMS_TOPK_TRAIN": 2000,
                "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN",
                    "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256,
                    "STEM_OUT_CHANNELS": 64, "DEFORM_ON_PER_STAGE": [False, False, False, False],
                    "DEFORM_MODULATED": False, "DEFORM_NUM_GROUPS": 1},
        "ROI_HEADS": {"NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"], "IOU_THRESHOLDS": [0.5],
                      "IOU_LABELS": [0, 1], "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25,
                      "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5, "PROPOSAL_APPEND_GT": True},
        "ROI_BOX_HEAD": {"NAME": "", "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                         "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "SMOOTH_L1_BETA": 0.0, "POOLER_RESOLUTION": 14,
                         "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "NUM_FC": 0, "FC_DIM": 1024,
                         "NUM_CONV": 0, "CONV_DIM": 256, "NORM": "", "CLS_AGNOSTIC_BBOX_REG": False,
                         "TRAIN_ON_PRED_BOXES": False},
        "ROI_MASK_HEAD": {"NAME": "MaskRCNNConvUpsampleHead", "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0,
                          "NUM_CONV": 0, "CONV_DIM": 256, "NORM": "", "CLS_AGNOSTIC_MASK": False,
                          "POOLER_TYPE": "ROIAlignV2"},
        "SEM_SEG_HEAD": {"NAME": "SemSegFPNHead", "IN_FEATURES": ["p2", "p3", "p4", "p5"], "IGNORE_VALUE": 255,
                         "NUM_CLASSES": 54, "CONVS_DIM": 128, "COMMON_STRIDE": 4, "NORM": "GN", "LOSS_WEIGHT": 1.0},
        # detectron2/config/defaults.py:398-406 (combine_semantic_and_instance_outputs thresholds)
        "PANOPTIC_FPN": {"INSTANCE_LOSS_WEIGHT": 1.0,
                         "COMBINE": {"ENABLED": True, "OVERLAP_THRESH": 0.5, "STUFF_AREA_LIMIT": 4096,
                                     "INSTANCES_CONFIDENCE_THRESH": 0.5}},
    },
}

_WSL_DEFAULTS = {
    "WSL": {"VIS_TEST": False, "ITER_SIZE": 1, "MEAN_LOSS": True, "SIZE_EPOCH": 5000, "CMIL": False, "USE_OBN": True,
            "REFINE_NUM": 3, "REFINE_REG": [False, False, False], "HAS_GAM": False, "REFINE_MIST": False,
            "CLS_AGNOSTIC_BBOX_KNOWN": False, "SAMPLING": {"SAMPLING_ON": False}, "CASCADE_ON": False, "PS_ON": False,
            "SP_ON": False, "MASK_MINED_TOP_K": 10},
    "MODEL": {"ROI_BOX_HEAD": {"DAN_DIM": [4096, 4096]},
              "SEM_SEG_HEAD": {"MASK_SOFTMAX": False, "CONSTRAINT": False}},
}

_C = CN(_DEFAULTS)


def add_wsl_config(cfg):
    """Add the project keys of projects/WSL (add_wsl_config there) to a config node in place."""
    cfg.merge_from_other_cfg(CN(_WSL_DEFAULTS))
