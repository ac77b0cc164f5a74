"""`cfg` container — same public behaviour as the reference's SegmentronConfig
(segmentron/config/config.py:13-127): dotted attribute access that auto-creates nested nodes,
string values literal_eval'ed, unknown keys rejected on update, freeze that stamps TIME_STAMP and
drops the MODEL.<OTHER_MODEL> sub-trees.  Re-implemented, not copied."""
import ast
import time

import yaml


class SegmentronConfig(dict):
    _FLAG = "immutable"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__[self._FLAG] = False

    # ------------------------------------------------------------------ attribute protocol
    def _node(self, key, create):
        if key not in self:
            if not create:
                raise KeyError(key)
            dict.__setitem__(self, key, SegmentronConfig())
        return self[key]

    def __getattr__(self, key, create_if_not_exist=True):
        if key == self._FLAG:
            return self.__dict__.setdefault(self._FLAG, False)
        return self._node(key, create_if_not_exist)

    def __setattr__(self, key, value, create_if_not_exist=True):
        if key == self._FLAG:
            self.__dict__[key] = value
            return
        *path, leaf = key.split(".")
        node = self
        for part in path:
            node = node.__getattr__(part, create_if_not_exist)
        if leaf not in node and not create_if_not_exist:
            raise KeyError(leaf)
        node[leaf] = value

    def __setitem__(self, key, value):
        if self.immutable:
            raise AttributeError(
                'Attempted to set "{}" to "{}", but SegConfig is immutable'.format(key, value))
        if isinstance(value, str):
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                pass
        super().__setitem__(key, value)

    # ------------------------------------------------------------------ updates
    def update_from_other_cfg(self, other):
        stack = [("", dict(other))]
        while stack:
            prefix, node = stack.pop(0)
            for k, v in node.items():
                full = prefix + "." + k if prefix else k
                if isinstance(v, dict):
                    stack.append((full, v))
                    continue
                try:
                    self.__setattr__(full, v, create_if_not_exist=False)
                except KeyError:
                    raise KeyError("Non-existent config key: {}".format(full))

    def update_from_list(self, config_list):
        if len(config_list) % 2:
            raise ValueError("Command line options config format error! Please check it: {}"
                             .format(config_list))
        for k, v in zip(config_list[0::2], config_list[1::2]):
            try:
                self.__setattr__(k, v, create_if_not_exist=False)
            except KeyError:
                raise KeyError("Non-existent config key: {}".format(k))

    def update_from_file(self, config_file):
        with open(config_file, "r", encoding="utf-8") as f:
            self.update_from_other_cfg(yaml.load(f, Loader=yaml.FullLoader))

    # ------------------------------------------------------------------ freeze
    def remove_irrelevant_cfg(self):
        from ..models.model_zoo import MODEL_REGISTRY
        name = self.MODEL.MODEL_NAME.lower()
        registered = [m.lower() for m in MODEL_REGISTRY.get_list()]
        assert name in registered, "Expected model name in {}, but received {}".format(
            MODEL_REGISTRY.get_list(), self.MODEL.MODEL_NAME)
        known = registered + _REFERENCE_ONLY_MODELS
        keep = {name}
        if name == "pointrend":
            keep.add(self.MODEL.POINTREND.BASEMODEL.lower())
        for key in [k for k in self.MODEL.keys() if k.lower() in known and k.lower() not in keep]:
            self.MODEL.pop(key)

    def check_and_freeze(self):
        self.TIME_STAMP = time.strftime("%Y-%m-%d-%H-%M", time.localtime())
        self.remove_irrelevant_cfg()
        self.immutable = True

    def set_immutable(self, immutable):
        self.immutable = immutable
        for v in self.values():
            if isinstance(v, SegmentronConfig):
                v.set_immutable(immutable)

    def is_immutable(self):
        return self.immutable


# model heads that own a MODEL.<NAME> config sub-tree in the reference but are outside the
# MI355X hot path; listed so their sub-trees are still dropped on freeze
# (CCNet is not registered in the reference either — models/__init__.py:11 — so MODEL.CCNET stays).
_REFERENCE_ONLY_MODELS = ["danet", "ocnet", "encnet", "cgnet", "pointrend", "hrnet"]
