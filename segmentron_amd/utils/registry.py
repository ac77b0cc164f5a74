"""name -> object registry with the reference's interface
(segmentron/utils/registry.py:48-78: register as decorator or call, get raises KeyError)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _add(self, name, obj):
        if name in self._obj_map:
            raise AssertionError("An object named '{}' was already registered in '{}' registry!"
                                 .format(name, self._name))
        self._obj_map[name] = obj

    def register(self, obj=None, name=None):
        if obj is not None:
            self._add(name or obj.__name__, obj)
            return None

        def deco(target):
            self._add(name or target.__name__, target)
            return target
        return deco

    def get(self, name):
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))

    def get_list(self):
        return list(self._obj_map)
