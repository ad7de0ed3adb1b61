"""detectron2.structures plumbing: Boxes, Instances, ImageList, BoxMode.

Containers only.  They carry the per-image fields of the batched-input schema
(/root/reference/cubercnn/data/dataset_mapper.py:133-155) across the registry boundary.  The
hot-path IoU / matching / decode arithmetic does NOT go through these helpers -- it runs in the
HIP kernels (omni3d_amd/csrc); `pairwise_iou` / `pairwise_ioa` here exist for API parity with
`detectron2.structures` and dispatch to the kernel on GPU tensors.
"""
import itertools
from enum import IntEnum, unique
from typing import Any, Dict, List, Tuple, Union

import torch


@unique
class BoxMode(IntEnum):
    XYXY_ABS = 0
    XYWH_ABS = 1
    XYXY_REL = 2
    XYWH_REL = 3
    XYWHA_ABS = 4

    @staticmethod
    def convert(box, from_mode, to_mode):
        if from_mode == to_mode:
            return box
        original_type = type(box)
        is_numpy = not isinstance(box, (torch.Tensor, list, tuple))
        single = isinstance(box, (list, tuple))
        arr = torch.tensor(box, dtype=torch.float64)[None, :] if single else (
            torch.from_numpy(__import__("numpy").asarray(box)).clone() if is_numpy else box.clone())
        if from_mode == BoxMode.XYWH_ABS and to_mode == BoxMode.XYXY_ABS:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        elif from_mode == BoxMode.XYXY_ABS and to_mode == BoxMode.XYWH_ABS:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        else:
            raise NotImplementedError(f"Conversion from BoxMode {from_mode} to {to_mode} is not supported")
        if single:
            return original_type(arr.flatten().tolist())
        if is_numpy:
            return arr.numpy()
        return arr


class Boxes:
    """(N, 4) XYXY absolute boxes."""

    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32, device=torch.device("cpu"))
        else:
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device=device))

    def area(self):
        box = self.tensor
        return (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold=0.0):
        box = self.tensor
        widths = box[:, 2] - box[:, 0]
        heights = box[:, 3] - box[:, 1]
        return (widths > threshold) & (heights > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on Boxes with {item} failed to return a matrix!"
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def inside_box(self, box_size, boundary_threshold=0):
        height, width = box_size
        return ((self.tensor[..., 0] >= -boundary_threshold) & (self.tensor[..., 1] >= -boundary_threshold)
                & (self.tensor[..., 2] < width + boundary_threshold) & (self.tensor[..., 3] < height + boundary_threshold))

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        assert all(isinstance(box, Boxes) for box in boxes_list)
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1: Boxes, boxes2: Boxes):
    """(N,4) x (M,4) -> (N,M) IoU; `inter > 0 ? inter / (a1 + a2 - inter) : 0` [detectron2]."""
    from ..kernels import det as kboxes
    return kboxes.pairwise_iou(boxes1.tensor, boxes2.tensor, mode="iou")


def pairwise_ioa(boxes1: Boxes, boxes2: Boxes):
    """(N,4) x (M,4) -> (N,M) intersection over area(boxes2) [detectron2]."""
    from ..kernels import det as kboxes
    return kboxes.pairwise_iou(boxes1.tensor, boxes2.tensor, mode="ioa")


class Instances:
    """Per-image bag of equally-long fields (detectron2.structures.Instances)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, f"Adding a field of length {data_len} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]):
        assert all(isinstance(i, Instances) for i in instance_lists)
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret

    def __str__(self):
        s = self.__class__.__name__ + "("
        s += f"num_instances={len(self)}, image_height={self._image_size[0]}, image_width={self._image_size[1]}, "
        s += "fields=[{}])".format(", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s

    __repr__ = __str__


class ImageList:
    """Batch tensor (N, C, H, W) zero-padded to a common size + the unpadded (h, w) per image."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0):
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            stride = size_divisibility
            max_h = (max_h + (stride - 1)) // stride * stride
            max_w = (max_w + (stride - 1)) // stride * stride
        batch_shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [max_h, max_w]
        batched = tensors[0].new_full(batch_shape, pad_value)
        for img, pad_img in zip(tensors, batched):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
