"""`DatasetMapper3D` (reference cubercnn/data/dataset_mapper.py:17-155): dataset dict -> model input dict.

Same steps and field names as the reference: read the image (BGR uint8), ResizeShortestEdge (+ RandomFlip when training),
transform the annotations (2D box hull, projected 3D centre, keypoints, pose mirroring under a flip, :76-129), build the
`Instances` (gt_classes, gt_boxes, gt_boxes3D = [u, v, z, w, h, l, X, Y, Z], gt_poses, gt_keypoints,
gt_unknown_category_mask, :133-155).

MI355X design: the pixel work -- PIL-bilinear resize and mirror -- runs in csrc/resize.hip (bit-exact
with the PIL call the reference makes) when the mapper is given a device, so the loader hands uint8 images to the GPU at
their ORIGINAL size and the multi-scale augmentation costs no host time; the annotation arithmetic is a few dozen floats
per image and stays on the host."""
import copy

import numpy as np
import torch

from ...d2.data import HFlipTransform, NoOpTransform, ResizeTransform, TransformList, resize_shortest_edge_size
from ...d2.structures import Boxes, BoxMode, Instances

_M1 = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]])             # dataset_mapper.py:63-72
_M2 = np.array([[-1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, 1.0]])


class Keypoints:
    """detectron2.structures.Keypoints: (N, K, 3) tensor wrapper"""

    def __init__(self, keypoints):
        self.tensor = torch.as_tensor(keypoints, dtype=torch.float32)

    def __len__(self):
        return self.tensor.size(0)

    def to(self, *args, **kwargs):
        return Keypoints(self.tensor.to(*args, **kwargs))

    def __getitem__(self, item):
        return Keypoints(self.tensor[[item]] if isinstance(item, int) else self.tensor[item])


def read_image(dataset_dict, fmt="BGR"):
    """detection_utils.read_image; synthetic datasets carry the pixels in memory (`image_array`, HWC uint8 in `fmt`)"""
    if "image_array" in dataset_dict:
        return np.asarray(dataset_dict["image_array"])
    from PIL import Image
    with Image.open(dataset_dict["file_name"]) as im:
        rgb = np.asarray(im.convert("RGB"))
    return rgb[:, :, ::-1] if fmt == "BGR" else rgb


class DatasetMapper3D:
    def __init__(self, cfg, is_train=True, device=None, rng=None):
        self.is_train = is_train
        self.image_format = cfg.INPUT.FORMAT
        if is_train:
            self.min_size, self.max_size, self.sample_style = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING
            self.flip = cfg.INPUT.RANDOM_FLIP
        else:
            self.min_size, self.max_size, self.sample_style, self.flip = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice", "none"
        if isinstance(self.min_size, int):
            self.min_size = (self.min_size,)
        if self.sample_style == "range":
            assert len(self.min_size) == 2, "more than 2 ({}) min_size(s) are provided for ranges".format(len(self.min_size))
        self.device = device
        self.rng = rng if rng is not None else np.random
        self.dataset_id_to_unknown_cats = {}

    # -- detectron2 build_augmentation(cfg, is_train): ResizeShortestEdge [+ RandomFlip(horizontal)] ----------------------
    def sample_transforms(self, h, w):
        nh, nw = resize_shortest_edge_size(h, w, self.min_size, self.max_size, self.sample_style, self.rng)
        tfm = TransformList([ResizeTransform(h, w, nh, nw)])
        do_flip = self.is_train and self.flip == "horizontal" and self.rng.uniform() < 0.5
        tfm.append(HFlipTransform(nw) if do_flip else NoOpTransform())
        return tfm, (nh, nw), do_flip

    def apply_image(self, image_hwc, out_hw, flip):
        """-> uint8 (3, nh, nw) tensor.  device set: csrc/resize.hip; else PIL on the host (the reference's call)"""
        nh, nw = out_hw
        if self.device is not None:
            from ...kernels import resize
            chw = torch.from_numpy(np.ascontiguousarray(image_hwc.transpose(2, 0, 1))).to(self.device, non_blocking=True)
            return resize.resize_bilinear_u8(chw, nh, nw, flip)
        from PIL import Image
        out = np.asarray(Image.fromarray(np.ascontiguousarray(image_hwc)).resize((nw, nh), Image.BILINEAR))
        if flip:
            out = np.flip(out, axis=1)
        return torch.as_tensor(np.ascontiguousarray(out.transpose(2, 0, 1)))

    def __call__(self, dataset_dict):
        dataset_dict = copy.deepcopy(dataset_dict)
        image = read_image(dataset_dict, self.image_format)
        if (image.shape[0], image.shape[1]) != (dataset_dict.get("height", image.shape[0]), dataset_dict.get("width", image.shape[1])):
            raise ValueError("Mismatched image shape for image {}".format(dataset_dict.get("file_name", dataset_dict.get("image_id"))))
        transforms, image_shape, flip = self.sample_transforms(image.shape[0], image.shape[1])
        dataset_dict["image"] = self.apply_image(image, image_shape, flip)
        dataset_dict.pop("image_array", None)
        if not self.is_train:
            return dataset_dict
        if "annotations" in dataset_dict:
            K = np.array(dataset_dict["K"])
            unknown = self.dataset_id_to_unknown_cats[dataset_dict["dataset_id"]]
            annos = [transform_instance_annotations(obj, transforms, K=K) for obj in dataset_dict.pop("annotations")
                     if obj.get("iscrowd", 0) == 0]
            inst = annotations_to_instances(annos, image_shape, unknown)
            dataset_dict["instances"] = inst[inst.gt_boxes.nonempty()] if len(inst) else inst      # filter_empty_instances
        return dataset_dict


def transform_instance_annotations(annotation, transforms, *, K):
    """dataset_mapper.py:76-129"""
    if isinstance(transforms, (tuple, list)) and not isinstance(transforms, TransformList):
        transforms = TransformList(transforms)
    bbox = BoxMode.convert(annotation["bbox"], annotation["bbox_mode"], BoxMode.XYXY_ABS)
    annotation["bbox"] = transforms.apply_box(np.array([bbox]))[0]
    annotation["bbox_mode"] = BoxMode.XYXY_ABS
    if annotation["center_cam"][2] != 0:
        point2D = K @ np.array(annotation["center_cam"])
        point2D[:2] = point2D[:2] / point2D[-1]
        annotation["center_cam_proj"] = point2D.tolist()
        annotation["center_cam_proj"][0:2] = transforms.apply_coords(point2D[np.newaxis][:, :2])[0].tolist()
        keypoints = (K @ np.array(annotation["bbox3D_cam"]).T).T
        keypoints[:, 0] /= keypoints[:, -1]
        keypoints[:, 1] /= keypoints[:, -1]
        keypoints[:, 2] = 1 if annotation["ignore"] else 2        # 0 unknown, 1 not visible, 2 visible (:103-113)
        transforms.apply_coords(keypoints[:, :2])
        annotation["keypoints"] = keypoints.tolist()
        for t in transforms:
            if isinstance(t, HFlipTransform):                      # manual mirror of the pose (:120-127)
                pose = _M1 @ np.array(annotation["pose"]) @ _M2
                annotation["pose"] = pose.tolist()
                annotation["R_cam"] = pose.tolist()
    return annotation


def annotations_to_instances(annos, image_size, unknown_categories):
    """dataset_mapper.py:133-155"""
    target = Instances(image_size)
    target.gt_classes = torch.tensor([int(obj["category_id"]) for obj in annos], dtype=torch.int64)
    target.gt_boxes = Boxes(torch.tensor(np.asarray([BoxMode.convert(obj["bbox"], obj["bbox_mode"], BoxMode.XYXY_ABS) for obj in annos],
                                                    dtype=np.float32).reshape(-1, 4)))
    target.gt_boxes3D = torch.FloatTensor([a["center_cam_proj"] + a["dimensions"] + a["center_cam"] for a in annos]).reshape(-1, 9)
    target.gt_poses = torch.FloatTensor([a["pose"] for a in annos]).reshape(-1, 3, 3)
    n = len(target.gt_classes)
    target.gt_keypoints = Keypoints(torch.FloatTensor([a["keypoints"] for a in annos]).reshape(-1, 8, 3))
    mask = torch.zeros(max(unknown_categories) + 1 if len(unknown_categories) else 1, dtype=bool)
    if len(unknown_categories):
        mask[torch.tensor(list(unknown_categories))] = True
    target.gt_unknown_category_mask = mask.unsqueeze(0).repeat([n, 1])
    return target
