import sys, torch, numpy as np
sys.path.insert(0, ".")
from oracle import backbone_oracle as bo
from centroids_reid_amd import backbone as bb
x = bo.synthetic_images(16, 128, 64, seed=4).cuda()
for init in ("perturbed", "product"):
    feats = {}
    for training in (False, True):
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            torch.manual_seed(0)
            net = bb.ResNet(last_stride=1)
            if init == "perturbed":
                net.load_state_dict(bo.make_state_dict("resnet50", 1, seed=5))
            net = net.cuda()
            eng = bb.BackboneEngine(net, dt)
            _, f = eng.forward(x, training)
            feats[(training, dt)] = f.clone()
        f32 = feats[(training, torch.float32)]
        for dt in (torch.bfloat16, torch.float16):
            r = ((feats[(training, dt)] - f32).norm(dim=1) / f32.norm(dim=1))
            print(init, "train" if training else "eval", dt, "rel err max %.2e mean %.2e" % (r.max().item(), r.mean().item()), "|f32| %.2f" % f32.norm(dim=1).mean().item())
