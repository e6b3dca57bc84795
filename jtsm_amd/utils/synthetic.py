"""Synthetic JTSM inputs in the reference's list[dict] contract:
per image a 3xSxS uint8-range float image, R proposals (log-uniform sizes, uniform corners) with
objectness in [0,1), block-grid superpixels of sp_block px with oh_labels[r,s] = 1 iff the centre of
block s lies in box r, `n_things` thing classes, `n_stuff` stuff bands in the semantic map (0 = things,
255 = ignore on the top rows).  Seeded per rank by the caller (1234 + rank)."""
import math

import torch

from ..structures import Boxes, Instances

NUM_THINGS, NUM_STUFF = 80, 54


def synthetic_inputs(seed, batch=2, size=1024, proposals=2000, sp_block=32, n_things=3, n_stuff=2, device="cpu",
                     num_things=NUM_THINGS, num_stuff=NUM_STUFF):
    g = torch.Generator().manual_seed(seed)
    grid = size // sp_block
    ids = ((torch.arange(size)[:, None] // sp_block) * grid + (torch.arange(size)[None, :] // sp_block)).to(torch.int32)
    centres = torch.arange(grid) * sp_block + sp_block / 2.0
    out = []
    for _ in range(batch):
        image = torch.rand(3, size, size, generator=g) * 255
        x0 = torch.rand(proposals, generator=g) * size * 0.75
        y0 = torch.rand(proposals, generator=g) * size * 0.75
        lo, hi = math.log(16.0), math.log(size / 2.0)
        w = torch.exp(torch.rand(proposals, generator=g) * (hi - lo) + lo)
        h = torch.exp(torch.rand(proposals, generator=g) * (hi - lo) + lo)
        boxes = torch.stack([x0, y0, (x0 + w).clamp(max=size), (y0 + h).clamp(max=size)], 1)
        objectness = torch.rand(proposals, generator=g)
        iny = (centres[None, :] >= boxes[:, 1:2]) & (centres[None, :] <= boxes[:, 3:4])
        inx = (centres[None, :] >= boxes[:, 0:1]) & (centres[None, :] <= boxes[:, 2:3])
        oh = (iny[:, :, None] & inx[:, None, :]).reshape(proposals, -1).to(torch.int32)
        things = torch.randperm(num_things, generator=g)[:n_things].sort().values
        stuff = torch.randperm(num_stuff - 1, generator=g)[:min(n_stuff, num_stuff - 1)] + 1
        sem = torch.zeros(size, size, dtype=torch.int64)
        band = size // (len(stuff) + 1)
        for j, s in enumerate(stuff):
            sem[(j + 1) * band:(j + 2) * band] = s
        sem[:8] = 255
        d = {
            "image": image.to(device),
            "instances": Instances((size, size), gt_classes=things.to(device)),
            "sem_seg": sem.to(device),
            "superpixels": ids.to(device),
            "proposals": Instances((size, size), proposal_boxes=Boxes(boxes.to(device)),
                                   objectness_logits=objectness.to(device), oh_labels=oh.to(device)),
        }
        out.append(d)
    return out
