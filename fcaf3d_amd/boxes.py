"""The members of the reference's GT box container that the FCAF3D path touches
(mmdet3d/core/bbox/structures/base_box3d.py:36-70, depth_box3d.py:41-48) and `bbox3d2result`
(mmdet3d/core/bbox/transforms.py:49-75)."""
import torch


class DepthInstance3DBoxes:
    """(m,7) boxes [x, y, z_bottom, dx, dy, dz, yaw]; `origin` = relative position of (x,y,z) inside the
    box of the tensor passed in ((.5,.5,0) = bottom centre, (.5,.5,.5) = gravity centre)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim))
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            self.box_dim = box_dim + 1
            self.with_yaw = False
        else:
            self.box_dim = box_dim
            self.with_yaw = with_yaw
        self.tensor = tensor.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            src = self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def bottom_center(self):
        return self.tensor[:, :3]

    @property
    def gravity_center(self):
        gc = self.tensor[:, :3].clone()
        gc[:, 2] = gc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    def to(self, device):
        out = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        out.tensor = self.tensor.to(device)
        out.box_dim, out.with_yaw = self.box_dim, self.with_yaw
        return out

    def __len__(self):
        return self.tensor.shape[0]


def bbox3d2result(bboxes, scores, labels):
    return dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())


def bbox3d2result_batch(bbox_list):
    """bbox3d2result for every scene of a batch with THREE device->host copies instead of three per scene
    (mmdet3d/core/bbox/transforms.py bbox3d2result semantics: boxes / scores / labels on the CPU)."""
    if not bbox_list:
        return []
    sizes = [len(s) for _, s, _ in bbox_list]
    boxes = torch.cat([b.tensor for b, _, _ in bbox_list]).cpu()
    scores = torch.cat([s for _, s, _ in bbox_list]).cpu()
    labels = torch.cat([l for _, _, l in bbox_list]).cpu()
    out, o = [], 0
    for (b, _, _), n in zip(bbox_list, sizes):
        cb = b.__class__.__new__(b.__class__)
        cb.tensor = boxes[o:o + n]
        cb.box_dim, cb.with_yaw = b.box_dim, b.with_yaw
        out.append(dict(boxes_3d=cb, scores_3d=scores[o:o + n], labels_3d=labels[o:o + n]))
        o += n
    return out
