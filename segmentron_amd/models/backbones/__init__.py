from .build import BACKBONE_REGISTRY, get_segmentation_backbone  # noqa: F401
from .xception import *  # noqa: F401,F403
from .resnet import *  # noqa: F401,F403
from .mobilenet import *  # noqa: F401,F403
from .hrnet import *  # noqa: F401,F403
