"""Default configuration tree — key names and default values are the drop-in contract with the
reference's yaml files (segmentron/config/settings.py:1-213); expressed as one nested literal."""
from .config import SegmentronConfig

_HR_STAGE = lambda mods, br, blocks, ch, block: dict(  # noqa: E731
    NUM_MODULES=mods, NUM_BRANCHES=br, NUM_BLOCKS=blocks, NUM_CHANNELS=ch, BLOCK=block,
    FUSE_METHOD="SUM")

DEFAULTS = dict(
    SEED=1024, TIME_STAMP="", ROOT_PATH="", PHASE="train",
    DATASET=dict(NAME="", MEAN=[0.5, 0.5, 0.5], STD=[0.5, 0.5, 0.5], IGNORE_INDEX=-1, WORKERS=4,
                 MODE="testval"),
    AUG=dict(MIRROR=True, BLUR_PROB=0.0, BLUR_RADIUS=0.0, COLOR_JITTER=None),
    TRAIN=dict(EPOCHS=30, BATCH_SIZE=1, CROP_SIZE=769, BASE_SIZE=1024,
               MODEL_SAVE_DIR="runs/checkpoints/", LOG_SAVE_DIR="runs/logs/",
               PRETRAINED_MODEL_PATH="", BACKBONE_PRETRAINED=True, BACKBONE_PRETRAINED_PATH="",
               RESUME_MODEL_PATH="", SYNC_BATCH_NORM=True, SNAPSHOT_EPOCH=10),
    SOLVER=dict(LR=1e-4, OPTIMIZER="sgd", EPSILON=1e-8, MOMENTUM=0.9, WEIGHT_DECAY=1e-4,
                DECODER_LR_FACTOR=10.0, LR_SCHEDULER="poly", POLY=dict(POWER=0.9),
                STEP=dict(GAMMA=0.1, DECAY_EPOCH=[10, 20]),
                WARMUP=dict(EPOCHS=0.0, FACTOR=1.0 / 3, METHOD="linear"),
                OHEM=False, AUX=False, AUX_WEIGHT=0.4, LOSS_NAME=""),
    TEST=dict(TEST_MODEL_PATH="", BATCH_SIZE=1, CROP_SIZE=None, SCALES=[1.0], FLIP=False),
    VISUAL=dict(OUTPUT_DIR="../runs/visual/"),
    MODEL=dict(
        MODEL_NAME="", BACKBONE="", BACKBONE_SCALE=1.0, MULTI_LOSS_WEIGHT=[1.0],
        DEFAULT_GROUP_NUMBER=32, DEFAULT_EPSILON=1e-5, BN_TYPE="BN", BN_EPS_FOR_ENCODER=None,
        BN_EPS_FOR_DECODER=None, OUTPUT_STRIDE=16, BN_MOMENTUM=None,
        DANET=dict(MULTI_DILATION=None, MULTI_GRID=False),
        DEEPLABV3_PLUS=dict(USE_ASPP=True, ENABLE_DECODER=True, ASPP_WITH_SEP_CONV=True,
                            DECODER_USE_SEP_CONV=True),
        OCNet=dict(OC_ARCH="base"),
        ENCNET=dict(SE_LOSS=True, SE_WEIGHT=0.2, LATERAL=True),
        CCNET=dict(RECURRENCE=2),
        CGNET=dict(STAGE2_BLOCK_NUM=3, STAGE3_BLOCK_NUM=21),
        POINTREND=dict(BASEMODEL="DeepLabV3_Plus"),
        HRNET=dict(PRETRAINED_LAYERS=["*"], STEM_INPLANES=64, FINAL_CONV_KERNEL=1, WITH_HEAD=True,
                   STAGE1=_HR_STAGE(1, 1, [1], [32], "BOTTLENECK"),
                   STAGE2=_HR_STAGE(1, 2, [4, 4], [32, 64], "BASIC"),
                   STAGE3=_HR_STAGE(1, 3, [4, 4, 4], [32, 64, 128], "BASIC"),
                   STAGE4=_HR_STAGE(1, 4, [4, 4, 4, 4], [32, 64, 128, 256], "BASIC")),
    ),
)


def _fill(node, tree):
    for k, v in tree.items():
        if isinstance(v, dict):
            _fill(node.__getattr__(k), v)
        else:
            node[k] = v


def make_default_cfg():
    c = SegmentronConfig()
    _fill(c, DEFAULTS)
    return c


cfg = make_default_cfg()


def reset_cfg():
    """Restore defaults in place (the reference's cfg is a freeze-once singleton; tests and
    bench.py need to build more than one model per process)."""
    cfg.set_immutable(False)
    for k in list(cfg.keys()):
        dict.__delitem__(cfg, k)
    _fill(cfg, DEFAULTS)
    return cfg
