"""detectron2.data plumbing: DatasetCatalog / MetadataCatalog registries and the two transforms the reference's mapper uses
(ResizeShortestEdge, RandomFlip) with detectron2's parameter sampling and coordinate rules."""
import types

import numpy as np


class _DatasetCatalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, "Dataset '{}' is already registered!".format(name)
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(name, ", ".join(self.keys()))) from e
        return f()

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


class Metadata(types.SimpleNamespace):
    """detectron2.data.catalog.Metadata: attribute bag with `.set(**kw)` / `.get(key, default)`; unknown attributes raise."""

    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)

    def as_dict(self):
        return dict(vars(self))


class _MetadataCatalog(dict):
    def get(self, name):
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]

    def list(self):
        return list(self.keys())


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()


# ---- transforms (detectron2.data.transforms) ---------------------------------------------------------------------------
class ResizeTransform:
    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class HFlipTransform:
    def __init__(self, width):
        self.width = width

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class NoOpTransform:
    def apply_coords(self, coords):
        return coords


class TransformList(list):
    def apply_coords(self, coords):
        for t in self:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        """detectron2 Transform.apply_box: transform the four corners, take the axis-aligned hull"""
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box, dtype=np.float64).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)


def resize_shortest_edge_size(h, w, short_edge_length, max_size, sample_style="range", rng=np.random):
    """detectron2 ResizeShortestEdge.get_transform / get_output_shape -> (new_h, new_w)"""
    if sample_style == "range":
        size = rng.randint(short_edge_length[0], short_edge_length[1] + 1)
    else:
        size = rng.choice(short_edge_length)
    if size == 0:
        return h, w
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)
