"""Compose / ToTensor / Normalize / ColorJitter with torchvision's semantics for the inputs the
reference feeds them (PIL RGB images or HWC uint8 arrays -> CHW float32 in [0, 1])."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div_(255.0)
        return t.to(torch.float32)


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std, self.inplace = mean, std, inplace

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        if not self.inplace:
            t = t.clone()
        return t.sub_(mean).div_(std)


class ColorJitter:
    """brightness / contrast / saturation jitter on PIL images (seg_data_base.py:42-45);
    hue jitter is not provided by this stand-in."""

    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        if hue:
            raise NotImplementedError("ColorJitter(hue) needs the real torchvision")
        self.ranges = [self._range(v) for v in (brightness, contrast, saturation)]

    @staticmethod
    def _range(v):
        if isinstance(v, (tuple, list)):
            return float(v[0]), float(v[1])
        return (max(0.0, 1.0 - v), 1.0 + v) if v else None

    def __call__(self, img):
        from PIL import ImageEnhance
        import random
        ops = [(ImageEnhance.Brightness, self.ranges[0]), (ImageEnhance.Contrast, self.ranges[1]),
               (ImageEnhance.Color, self.ranges[2])]
        random.shuffle(ops)
        for enh, rng in ops:
            if rng is not None:
                img = enh(img).enhance(random.uniform(*rng))
        return img
