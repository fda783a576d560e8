import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import backbone_oracle as bo
from centroids_reid_amd import backbone as bb
x = bo.synthetic_images(4, 128, 64, seed=21).cuda()
coef = torch.from_numpy(np.random.default_rng(3).standard_normal((4, 2048)).astype(np.float32)).cuda()
grads = []
for fuse in (True, False):
    sd = bo.make_state_dict('resnet50', 1, seed=1234)
    net = bb.ResNet(last_stride=1); net.load_state_dict(sd); net = net.cuda()
    eng = bb.BackboneEngine(net, torch.bfloat16); eng.fuse_bn_reduce = fuse
    eng.forward(x, training=True); eng.backward(coef)
    grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
names = list(grads[0])
for n in reversed(names):
    a, b = grads[0][n].double().flatten(), grads[1][n].double().flatten()
    print(f"{n:36s} rel {float((a-b).norm()/(b.norm()+1e-30)):.3e}")
