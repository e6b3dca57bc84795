"""Synthetic JTSM inputs in the reference's list[dict] contract:
per image a 3xSxS uint8-range float image, R proposals (log-uniform sizes, uniform corners) with
objectness in [0,1), block-grid superpixels of sp_block px with oh_labels[r,s] = 1 iff the centre of
block s lies in box r, `n_things` thing classes, `n_stuff` stuff bands in the semantic map (0 = things,
255 = ignore on the top rows).  Seeded per rank by the caller (1234 + rank).

`width` (default: `size`) makes the images size x width — BASELINE configs[4]'s Cityscapes shape is the same proposal
recipe scaled by width / size in x.  `cluster` > 0 replaces that fraction of the proposals by jittered copies of
`objects` (default `n_things`) "object" rectangles per image, the way real proposal sets (MCG, selective search) crowd
around regions: a mined pseudo-GT box that is one of those copies has O(cluster * R / objects) proposals above IoU 0.5,
which is what gives the mask branch a realistic foreground count (uniform-random boxes almost never overlap that
much).  With cluster = 1 every proposal belongs to a group, so the count no longer depends on WHICH proposal a
randomly initialised detector happens to pick."""
import math

import torch

from ..structures import Boxes, Instances

NUM_THINGS, NUM_STUFF = 80, 54


def synthetic_inputs(seed, batch=2, size=1024, proposals=2000, sp_block=32, n_things=3, n_stuff=2, device="cpu",
                     num_things=NUM_THINGS, num_stuff=NUM_STUFF, width=None, cluster=0.0, objects=None):
    g = torch.Generator().manual_seed(seed)
    width = size if width is None else width
    sx = width / float(size)
    gh, gw = size // sp_block, width // sp_block
    ids = ((torch.arange(size)[:, None] // sp_block) * gw + (torch.arange(width)[None, :] // sp_block)).to(torch.int32)
    cy = torch.arange(gh) * sp_block + sp_block / 2.0
    cx = torch.arange(gw) * sp_block + sp_block / 2.0
    out = []
    for _ in range(batch):
        image = torch.rand(3, size, width, generator=g) * 255
        x0 = torch.rand(proposals, generator=g) * size * 0.75
        y0 = torch.rand(proposals, generator=g) * size * 0.75
        lo, hi = math.log(16.0), math.log(size / 2.0)
        w = torch.exp(torch.rand(proposals, generator=g) * (hi - lo) + lo)
        h = torch.exp(torch.rand(proposals, generator=g) * (hi - lo) + lo)
        boxes = torch.stack([x0 * sx, y0, ((x0 + w) * sx).clamp(max=width), (y0 + h).clamp(max=size)], 1)
        objectness = torch.rand(proposals, generator=g)
        if cluster > 0:
            # (drawn after everything the plain recipe draws, so cluster = 0 reproduces it bit for bit)
            nc = int(round(cluster * proposals))
            no = n_things if objects is None else int(objects)
            ox = torch.rand(no, generator=g) * width * 0.6
            oy = torch.rand(no, generator=g) * size * 0.6
            if objects is None:
                ow = (torch.rand(no, generator=g) * 0.25 + 0.12) * width
                ohh = (torch.rand(no, generator=g) * 0.25 + 0.12) * size
            else:   # many groups: log-uniform sides like the plain recipe, so the groups span the FPN levels
                ow = torch.exp(torch.rand(no, generator=g) * (hi - lo) + lo) * sx
                ohh = torch.exp(torch.rand(no, generator=g) * (hi - lo) + lo)
            which = torch.randint(0, no, (nc,), generator=g)
            jit = (torch.rand(nc, 4, generator=g) - 0.5) * 0.24          # each edge moves by up to 12 % of the side
            obj = torch.stack([ox, oy, ox + ow, oy + ohh], 1)[which]
            side = torch.stack([ow, ohh, ow, ohh], 1)[which]
            cl = obj + jit * side
            cl[:, 0::2] = cl[:, 0::2].clamp(0, width)
            cl[:, 1::2] = cl[:, 1::2].clamp(0, size)
            boxes[:nc] = cl
        iny = (cy[None, :] >= boxes[:, 1:2]) & (cy[None, :] <= boxes[:, 3:4])
        inx = (cx[None, :] >= boxes[:, 0:1]) & (cx[None, :] <= boxes[:, 2:3])
        oh = (iny[:, :, None] & inx[:, None, :]).reshape(proposals, -1).to(torch.int32)
        things = torch.randperm(num_things, generator=g)[:n_things].sort().values
        stuff = torch.randperm(num_stuff - 1, generator=g)[:min(n_stuff, num_stuff - 1)] + 1
        sem = torch.zeros(size, width, dtype=torch.int64)
        band = size // (len(stuff) + 1)
        for j, s in enumerate(stuff):
            sem[(j + 1) * band:(j + 2) * band] = s
        sem[:8] = 255
        d = {
            "image": image.to(device),
            "instances": Instances((size, width), gt_classes=things.to(device)),
            "sem_seg": sem.to(device),
            "superpixels": ids.to(device),
            "proposals": Instances((size, width), proposal_boxes=Boxes(boxes.to(device)),
                                   objectness_logits=objectness.to(device), oh_labels=oh.to(device)),
        }
        out.append(d)
    return out
