"""Drop-in name: `import segmentron` resolves to the MI355X implementation, so the reference's
entry points (`from segmentron.config import cfg`, `from segmentron.models.model_zoo import
get_segmentation_model`, ...) bind to the HIP path unchanged.  Pure aliasing, no logic."""
import importlib
import sys

import segmentron_amd as _impl

_ALIASES = ["config", "config.config", "config.settings", "utils", "utils.registry", "modules",
            "modules.basic", "modules.module", "modules.batch_norm", "models", "models.model_zoo",
            "models.segbase", "models.deeplabv3_plus", "models.fcn", "models.pspnet", "models.backbones",
            "models.backbones.build", "models.backbones.xception", "models.backbones.resnet", "models.backbones.mobilenet", "models.backbones.hrnet", "models.hrnet_seg", "data", "data.dataloader"]
for _name in _ALIASES:
    sys.modules["segmentron." + _name] = importlib.import_module("segmentron_amd." + _name)
config, utils, modules, models, data = (sys.modules["segmentron." + n] for n in
                                        ("config", "utils", "modules", "models", "data"))
__all__ = ["config", "utils", "modules", "models", "data"]
