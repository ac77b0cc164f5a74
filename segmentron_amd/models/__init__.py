"""Model zoo (hot-path heads; names as registered in segmentron/models/__init__.py)."""
from .model_zoo import MODEL_REGISTRY, get_segmentation_model  # noqa: F401
from .deeplabv3_plus import DeepLabV3Plus  # noqa: F401
from .fcn import FCN  # noqa: F401
from .pspnet import PSPNet  # noqa: F401
from .hrnet_seg import HighResolutionNet  # noqa: F401
from .ccnet import CCNet  # noqa: F401
from .fast_scnn import FastSCNN  # noqa: F401
from .danet import DANet  # noqa: F401
