"""smoke(): one tiny CTLModel.training_step on cuda:0 (fp32 parity mode) checked against the CPU oracle."""
import numpy as np
import torch


def run():
    from oracle import backbone_oracle as bo, reid_oracle as ro
    from .bench_train import make_model
    P, K, C, H, W = 4, 4, 20, 64, 32
    model = make_model(num_classes=C, dtype=torch.float32, K=K)
    sd = bo.make_state_dict("resnet50", 1, seed=5)
    model.backbone.base.load_state_dict(sd)
    model.backbone.base.cuda()
    x = bo.synthetic_images(P * K, H, W, seed=3)
    labels = torch.as_tensor(np.repeat(np.arange(P) * 3 % C, K).astype(np.int64))
    is_real = torch.ones(P * K, dtype=torch.bool)
    centers0 = model.center_loss.centers.detach().cpu().clone(); fc0 = model.fc_query.weight.detach().cpu().clone()
    out = model.training_step((x.cuda(), labels.cuda(), torch.zeros(P * K, dtype=torch.int64), is_real), 0)
    torch.set_num_threads(16)
    with torch.no_grad():
        _, feat = bo.backbone_forward(x, {k: v.clone() for k, v in sd.items()}, "resnet50", 1, training=True)
        o = ro.ctl_heads(feat, labels, is_real, torch.ones(2048), torch.zeros(2048), torch.zeros(2048), torch.ones(2048),
                         fc0, centers0, P, K)
    got, ref = float(out["loss"]), float(o["total"])
    assert abs(got - ref) < 5e-4 * max(1.0, abs(ref)), (got, ref)
    print(f"smoke train step OK: loss {got:.6f} (oracle {ref:.6f})")
