"""OICROutputLayers — surface of projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:448-586 (layers,
losses), :684-783 (predict_probs / predict_boxes and their K-head averages), :616-646 (inference) and the
module-level fast_rcnn_inference / fast_rcnn_inference_single_image (:48-163).  `cls_score: Linear(in, K+1)`,
`bbox_pred: Linear(in, 4K)`; losses = instance-weighted CE and L1 (jtsm_amd/csrc/wsl_losses.hip); inference =
one fused predict launch + one per-image detection call (jtsm_amd/csrc/postprocess.hip)."""
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ...layers.postprocess import fast_rcnn_inference_device, oicr_predict
from ...layers.wrappers import Linear, cat
from ...layers.wsl_losses import oicr_loss
from ...structures import Boxes, Instances
from ..box_regression import Box2BoxTransform


_SIDE = {}


def _side_stream(device):
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st


def fast_rcnn_inference(boxes: List[torch.Tensor], scores: List[torch.Tensor], image_shapes: List[Tuple[int, int]],
                        score_thresh: float, nms_thresh: float, topk_per_image: int):
    """Per image: (Instances, kept proposal rows, all_scores (1,R,K+1), all_boxes (1,R,4K)) — as the reference.
    Every image's device call is launched before the first count is read back (the reference synchronises per image, in
    nonzero()); every second image's call goes to a side stream: the greedy scan of an image's candidates is one
    workgroup's serial walk, and two images' walks run side by side."""
    launched = []
    dev = boxes[0].device if len(boxes) else None
    side = _side_stream(dev) if (dev is not None and dev.type == "cuda" and len(boxes) > 1) else None
    main = torch.cuda.current_stream(dev) if side is not None else None
    if side is not None:
        side.wait_stream(main)
    for k, (s, b, shape) in enumerate(zip(scores, boxes, image_shapes)):
        if side is not None and k % 2 == 1:
            with torch.cuda.stream(side):
                launched.append(_launch_single_image(b, s, shape, score_thresh, nms_thresh, topk_per_image))
        else:
            launched.append(_launch_single_image(b, s, shape, score_thresh, nms_thresh, topk_per_image))
    if side is not None:
        main.wait_stream(side)
        for k in range(1, len(launched), 2):        # allocated under the side stream, read on the compute stream from here on
            for t in list(launched[k][0].values()) + [launched[k][1], launched[k][2]]:
                t.record_stream(main)
    per_image = [_finish_single_image(*x) for x in launched]
    return tuple([x[i] for x in per_image] for i in range(4))


@torch.no_grad()
def _launch_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    all_scores, all_boxes = scores.clone().unsqueeze(0), boxes.clone().unsqueeze(0)
    out = fast_rcnn_inference_device(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image)
    return out, all_scores, all_boxes, image_shape


@torch.no_grad()
def _finish_single_image(out, all_scores, all_boxes, image_shape):
    n = int(out["count"].item())          # the data-dependent length (the reference synchronises in nonzero())
    result = Instances(image_shape)
    result.pred_boxes = Boxes(out["boxes"][:n])
    result.scores = out["scores"][:n]
    result.pred_classes = out["classes"][:n]
    result.pred_inds = out["rows"][:n]
    return result, out["rows"][:n], all_scores, all_boxes


@torch.no_grad()
def fast_rcnn_inference_single_image(boxes, scores, image_shape: Tuple[int, int], score_thresh: float,
                                     nms_thresh: float, topk_per_image: int):
    """Score threshold, per-class NMS and top-k of one image, in one library call
    (jtsm_fast_rcnn_inference_f32).  `pred_inds` / the second return value are proposal rows of the INPUT (the
    reference's second value indexes the rows left after dropping non-finite predictions; identical when all
    predictions are finite)."""
    return _finish_single_image(*_launch_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image))


class OICROutputLayers(nn.Module):
    def __init__(self, input_size, *, num_classes, box2box_transform, refine_k, refine_reg, loss_weight=1.0,
                 test_score_thresh=0.0, test_nms_thresh=0.5, test_topk_per_image=100):
        super().__init__()
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.num_classes = num_classes
        self.box_dim = len(box2box_transform.weights)
        self.num_bbox_reg_classes = num_classes
        self.cls_score = Linear(input_size, num_classes + 1)
        self.bbox_pred = Linear(input_size, num_classes * self.box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.refine_k = refine_k
        self.refine_reg = refine_reg
        self.loss_weight = {"loss_box_reg": loss_weight} if isinstance(loss_weight, float) else loss_weight

    @classmethod
    def from_config(cls, cfg, input_size, refine_k):
        return cls(input_size, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                   box2box_transform=Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
                   refine_k=refine_k, refine_reg=cfg.WSL.REFINE_REG,
                   loss_weight={"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
                   test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                   test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
                   test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE)

    @property
    def has_reg(self):
        return bool(self.refine_reg[self.refine_k])

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        scores = self.cls_score(x)
        if self.has_reg:
            return scores, self.bbox_pred(x)
        return scores, torch.zeros(scores.shape[0], self.num_bbox_reg_classes * self.box_dim, dtype=scores.dtype,
                                   device=scores.device)

    def losses(self, predictions, proposals, gt_classes=None, gt_boxes=None, gt_weights=None):
        """predictions = (logits (R,K+1), deltas (R,4K)).  Two ways in:

        * the reference's `losses(predictions, proposals)` (fast_rcnn_oicr.py:562-586, OICROutputs.__init__ :180-250):
          `proposals` is a list[Instances] with `proposal_boxes`, `gt_classes`, `gt_weights` and (optionally, an image
          without targets has none: the proposal boxes stand in, :232-236) `gt_boxes`;
        * the fused form JTSMROIHeads uses: `proposals` is the (R,4) tensor of proposal boxes cat'ed over the images and
          the three label tensors are passed beside it (they come straight from the labelling kernel)."""
        scores, deltas = predictions
        if isinstance(proposals, (list, tuple)):
            if not len(proposals):
                zero = 0.0 * scores.sum()
                k = "_r" + str(self.refine_k)
                return {"loss_cls" + k: zero, **({"loss_box_reg" + k: 0.0 * deltas.sum()} if self.has_reg else {})}
            proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
            assert not proposal_boxes.requires_grad, "Proposals should not require gradients!"
            gt_classes = cat([p.gt_classes for p in proposals], dim=0)
            gt_boxes = cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
            gt_weights = cat([p.gt_weights for p in proposals], dim=0)
        else:
            proposal_boxes = proposals
        lc, lb = oicr_loss(scores, deltas if self.has_reg else None, gt_classes, gt_weights,
                           proposal_boxes if self.has_reg else None, gt_boxes if self.has_reg else None)
        k = "_r" + str(self.refine_k)
        out = {"loss_cls" + k: lc}
        if self.has_reg:
            out["loss_box_reg" + k] = lb * self.loss_weight.get("loss_box_reg", 1.0)
        return out

    def predict_probs(self, predictions, counts):
        return F.softmax(predictions[0], dim=-1).split(counts, dim=0)

    def predict_boxes(self, predictions, proposal_boxes, counts):
        return self.box2box_transform.apply_deltas(predictions[1], proposal_boxes).split(counts)

    def inference(self, predictions, proposals: List[Instances]):
        """`predictions`: (logits, deltas) of this head, or a list of such pairs — then probabilities and deltas
        are averaged over the heads (predict_probs_K / predict_boxes_K).  Returns what fast_rcnn_inference does."""
        heads = list(predictions) if isinstance(predictions[0], (tuple, list)) else [predictions]
        if not len(proposals):
            return [], [], [], []
        counts = [len(p) for p in proposals]
        prop = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        probs, boxes = oicr_predict([h[0] for h in heads], [h[1] for h in heads], prop,
                                    self.box2box_transform.weights, self.box2box_transform.scale_clamp)
        return fast_rcnn_inference(boxes.split(counts), probs.split(counts), [x.image_size for x in proposals],
                                   self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image)
