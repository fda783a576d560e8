import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import backbone_oracle as bo
from centroids_reid_amd import backbone as bb
torch.set_num_threads(32)
arch='resnet50'; B,H,W = 2, int(sys.argv[1]) if len(sys.argv)>1 else 128, int(sys.argv[2]) if len(sys.argv)>2 else 64
sd = bo.make_state_dict(arch, 1, seed=1234)
net = bb.ResNet(last_stride=1); net.load_state_dict(sd); net = net.cuda()
eng = bb.BackboneEngine(net, torch.float32)
x = bo.synthetic_images(B, H, W, seed=7)
coef = torch.from_numpy(np.random.default_rng(99).standard_normal((B, 2048)).astype(np.float32))
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(("running_mean","running_var"))}
sd2 = {**{k: v.clone() for k, v in sd.items()}, **params}
_, feat = bo.backbone_forward(x, sd2, arch, 1, training=True)
(feat*coef).sum().backward()
_, fg = eng.forward(x.cuda(), training=True)
print('feat maxdiff', (fg.cpu()-feat.detach()).abs().max().item())
eng.backward(coef.cuda())
for name, p in net.named_parameters():
    ref = params[name].grad.numpy(); got = p.grad.cpu().numpy()
    err = np.abs(got-ref).max()/ (np.abs(ref).max()+1e-30)
    bad = np.abs(got-ref) > 5e-3*np.abs(ref).max()
    flag = ''
    if bad.any():
        idx = np.argwhere(bad)
        flag = f' BAD {bad.sum()}/{bad.size} first {idx[:3].tolist()} rows~{np.unique(idx[:,0])[:8].tolist()} cols~{np.unique(idx[:,1])[:8].tolist() if idx.shape[1]>1 else ""}'
    print(f'{name:40s} relerr {err:.2e}{flag}')
