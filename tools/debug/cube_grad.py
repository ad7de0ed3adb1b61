"""debug: where does the cube-head backward of the HIP path leave the CPU oracle (full-size batch 4)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import make_golden as MG
from oracle import model_oracle as MO
from omni3d_amd import synthetic
import omni3d_amd.functional as HF

priors = synthetic.make_priors(50)
model = MG.build_product_model(MG.product_cfg([]), priors, 5, device="cpu")
oracle = MO.ModelOracle(priors)
oracle.load_state_dict(model.state_dict(), strict=True)
model = model.to("cuda")
batch = synthetic.make_batch(4, 512, 512, num_gt=8, seed=21, priors=priors)
B = 4
A = 3 * sum((512 // s) ** 2 for s in (4, 8, 16, 32, 64))
g = torch.Generator().manual_seed(3)
E_rpn, E_roi = torch.empty(B, A).exponential_(generator=g), torch.empty(B, 2048).exponential_(generator=g)
model.train(); oracle.train()
cap = {}
ch = oracle.roi_heads.cube_head
def hook(name):
    def f(mod, gin, gout):
        cap["o_" + name + "_gout"] = gout[0].detach().clone()
        cap["o_" + name + "_gin"] = gin[0].detach().clone() if gin[0] is not None else None
    return f
ch.feature_generator.fc2.register_full_backward_hook(hook("fc2"))
ch.feature_generator.fc1.register_full_backward_hook(hook("fc1"))
ch.bbox_3D_pose.register_full_backward_hook(hook("pose"))
ch.bbox_3D_dims.register_full_backward_hook(hook("dims"))
ch.bbox_3D_center_deltas.register_full_backward_hook(hook("xy"))
ch.bbox_3D_center_depth.register_full_backward_hook(hook("z"))
ch.bbox_3D_uncertainty.register_full_backward_hook(hook("unc"))
ref = oracle(batch, E_rpn, E_roi)
sum(ref.values()).backward()
model.proposal_generator.injected = {"E": E_rpn, "proposals": oracle.last_proposals}
model.roi_heads.injected = {"E": E_roi}
# capture HIP-side tensors
hch = model.roi_heads.cube_head
orig_fwd = hch.forward
def fwd(x):
    x.retain_grad(); cap["h_x"] = x
    h1 = hch.feature_generator.fc1(x, relu=True); h1.retain_grad(); cap["h_h1"] = h1
    h2 = hch.feature_generator.fc2(h1, relu=True); h2.retain_grad(); cap["h_h2"] = h2
    w, b = hch.fused_parameters()
    head = HF.linear(h2, w, b); head.retain_grad(); cap["h_head"] = head
    return head
hch.forward = fwd
losses = model(batch)
sum(losses.values()).backward()
cls = model.roi_heads.last_sampled_classes[:, :128].reshape(-1).cpu()
fg = (cls >= 0) & (cls < 50)
print("fg rows", int(fg.sum()), "oracle rows", cap["o_fc2_gout"].shape)
def cmp(name, a, b):
    a, b = a.double().cpu(), b.double().cpu()
    print(f"{name:28s} rel L2 {float((a-b).norm()/b.norm()):.3e}  max|d| {float((a-b).abs().max()):.3e} |b|max {float(b.abs().max()):.3e}")
    return a, b
# d(h2) : HIP grad of h2 (post-relu features) vs oracle grad_output of fc2 module is grad wrt fc2 output BEFORE relu... use relu'd convention
dh2_h = cap["h_h2"].grad[fg.cuda()]
# oracle: grad wrt f (post relu) = grad_in of the heads; approximate via sum; instead compare pre-activation grads:
# HIP: relu_bwd is inside _Linear.backward, so grad wrt fc2 *input* (h1) is comparable with oracle fc2 grad_input
cmp("d h1 (fc2 grad_input)", cap["h_h1"].grad[fg.cuda()], cap["o_fc2_gin"])
cmp("d x  (fc1 grad_input)", cap["h_x"].grad.permute(0, 2, 3, 1)[fg.cuda()].permute(0, 3, 1, 2).reshape(int(fg.sum()), -1) if False else cap["h_x"].grad[fg.cuda()].reshape(int(fg.sum()), -1), cap["o_fc1_gin"])
# head grads: pose columns
K = 50
dhead = cap["h_head"].grad[fg.cuda()]
pose_cols = dhead[:, 6 * K: 12 * K]
cmp("d pose out", pose_cols, cap["o_pose_gout"])
nz = (cap["h_head"].grad[~fg.cuda()].abs().sum())
print("grad mass on non-fg rows of head:", float(nz), " on fg rows:", float(dhead.abs().sum()))
print("grad mass on non-fg rows of h2:", float(cap["h_h2"].grad[~fg.cuda()].abs().sum()), " on fg rows:", float(dh2_h.abs().sum()))
print("grad mass on non-fg rows of h1:", float(cap["h_h1"].grad[~fg.cuda()].abs().sum()))
# per-row error of d h1
a = cap["h_h1"].grad[fg.cuda()].double().cpu(); b = cap["o_fc2_gin"].double()
rowerr = (a - b).norm(dim=1) / b.norm(dim=1).clamp(min=1e-30)
print("rows with rel err > 1e-3:", int((rowerr > 1e-3).sum()), "of", len(rowerr), "worst", rowerr.topk(5))

r = int(rowerr.argmax())
c = int(cls[fg][r])
print("row", r, "class", c)
segs = {"xy": (0, 2 * K, 2), "z": (2 * K, 3 * K, 1), "dims": (3 * K, 6 * K, 3), "pose": (6 * K, 12 * K, 6), "unc": (12 * K, 13 * K, 1)}
for name, (s0, s1, w) in segs.items():
    hh = dhead[r, s0:s1].double().cpu().view(K, w)[c]
    oo = cap["o_" + name + "_gout"][r].double().view(K, w)[c]
    print(name, "hip", hh.tolist(), "oracle", oo.tolist())
