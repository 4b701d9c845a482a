"""Value types passed across every module seam of the path, with the reference's semantics:
BoxList (mega_core/structures/bounding_box.py:9-266), cat_boxlist (structures/boxlist_ops.py:103-133),
ImageList / to_image_list (structures/image_list.py:7-72).  Only the xyxy mode and the methods the
inference path touches are provided.
"""
import torch


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(bbox.ndimension()))
        if bbox.size(-1) != 4:
            raise ValueError("last dimension of bbox should have a size of 4, got {}".format(bbox.size(-1)))
        if mode != "xyxy":
            raise ValueError("mode should be 'xyxy'")
        self.bbox = bbox
        self.size = image_size  # (image_width, image_height)
        self.mode = mode
        self.extra_fields = {}

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def convert(self, mode):
        if mode != "xyxy":
            raise ValueError("only xyxy is supported on the inference path")
        return self

    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def clip_to_image(self, remove_empty=True):
        """bounding_box.py:214-224 (in place, like the reference)."""
        self.bbox[:, 0].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 1].clamp_(min=0, max=self.size[1] - 1)
        self.bbox[:, 2].clamp_(min=0, max=self.size[0] - 1)
        self.bbox[:, 3].clamp_(min=0, max=self.size[1] - 1)
        if remove_empty:
            box = self.bbox
            keep = (box[:, 3] > box[:, 1]) & (box[:, 2] > box[:, 0])
            return self[keep]
        return self

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        if not isinstance(fields, (list, tuple)):
            fields = [fields]
        for f in fields:
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return out

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)


def cat_boxlist(bboxes):
    """boxlist_ops.py:103-133."""
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size = bboxes[0].size
    assert all(b.size == size for b in bboxes)
    fields = set(bboxes[0].fields())
    assert all(set(b.fields()) == fields for b in bboxes)
    out = BoxList(torch.cat([b.bbox for b in bboxes], dim=0), size, "xyxy")
    for f in fields:
        out.add_field(f, torch.cat([b.get_field(f) for b in bboxes], dim=0))
    return out


class ImageList(object):
    """image_list.py:7-27."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes  # list of (h, w)

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    """image_list.py:29-72 (SIZE_DIVISIBILITY = 0 on this path: no padding)."""
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        shapes = {tuple(t.shape) for t in tensors}
        if len(shapes) != 1:
            raise ValueError("frames of one clip must share a size (SIZE_DIVISIBILITY=0 path)")
        return ImageList(torch.stack(list(tensors)), [t.shape[-2:] for t in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
