"""Only what the model constructors touch: ``datasets[name].NUM_CLASS``
(segmentron/models/segbase.py:21; values from segmentron/data/dataloader/*.py `NUM_CLASS`).
The PIL datasets / augmentation pipeline are outside the hot path (SURVEY.md §2 OUT)."""


class _DatasetInfo:
    def __init__(self, name, num_class):
        self.NAME, self.NUM_CLASS = name, num_class


datasets = {
    "ade20k": _DatasetInfo("ade20k", 150),
    "pascal_voc": _DatasetInfo("pascal_voc", 21),
    "pascal_aug": _DatasetInfo("pascal_aug", 21),
    "coco": _DatasetInfo("coco", 21),
    "cityscape": _DatasetInfo("cityscape", 19),
    "sbu": _DatasetInfo("sbu", 2),
}
