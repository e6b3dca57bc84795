"""Training trajectories with and without the side streams: K optimizer steps from the same weights and seeds, every
loss of every step compared bit for bit (weights move every step, so every kernel sees fresh values each time).
usage: trajectory_soak.py [K] [REPEATS]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.layers import conv as K
from jtsm_amd.modeling.meta_arch import mcnn
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
init = copy.deepcopy(model.state_dict())
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
REPEATS = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def trajectory(sem_side, wgrad_side):
    mcnn.SEM_SIDE_STREAM, K.WGRAD_STREAM = sem_side, wgrad_side
    model.load_state_dict(init)
    opt = bench.make_optimizer(model)
    torch.manual_seed(7)
    out = []
    for _ in range(STEPS):
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        out.append(torch.stack([losses[k].detach() for k in sorted(losses)]))
    torch.cuda.synchronize()
    return torch.stack(out)


ref = trajectory(False, False)
print("one stream: final losses", [round(float(x), 5) for x in ref[-1]], flush=True)
again = trajectory(False, False)
print("one stream again: identical", bool(torch.equal(ref, again)), flush=True)
# The semantic head on its own stream sums the pyramid's gradients in another order (its term reaches the shared maps
# through autograd's accumulation, not through layers/grad_fan.py): its trajectories are compared with each other.
bad, firsts = 0, {}
for r in range(REPEATS):
    for sem, wg in ((True, True), (False, True), (True, False)):
        t = trajectory(sem, wg)
        want = ref if not sem else firsts.setdefault((sem, wg), t)
        same = bool(torch.equal(want, t))
        bad += not same
        first = int((want != t).any(dim=1).nonzero()[0]) if not same else -1
        print("repeat %d semantic side stream %d, weight-gradient side stream %d: identical to %s: %s%s; largest loss difference from one stream %.2e" % (
            r, sem, wg, "the one-stream run" if not sem else "the first such run", same, "" if same else " (first differing step %d)" % first,
            float((t - ref).abs().max())), flush=True)
print("trajectory soak: %d of %d trajectories differ from their reference" % (bad, REPEATS * 3))
