import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import util_models as um
from trackformer_amd import config, factory, fused, _cabi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for M, K, N in () if not fused.split_linear_enabled() else ((638, 256, 256), (638, 256, 384), (638, 256, 1024), (638, 1024, 256), (305, 256, 512), (22223, 256, 256)):
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) / 16).to(dev))
    b = torch.randn(N, generator=g).to(dev)
    ys = [fused.linear(x, w, b, relu=(N == 1024)).clone() for _ in range(4)]
    ref = (x.double() @ w.double().t() + b.double())
    if N == 1024: ref = ref.clamp_min(0)
    print("linear %dx%dx%d  run-to-run max diff %.3g, vs float64 %.3g" % (M, K, N, max(float((ys[0] - y).abs().max()) for y in ys[1:]), float((ys[0].double() - ref).abs().max())))
model, post, args = um.build("cfg2_deformable_tracking", factory.build_model, config.make_args, device=dev)
model.to(dev).tracking()
img = torch.randn(1, 3, 160, 192, generator=g).to(dev)
target = [{'track_query_hs_embeds': torch.randn(5, 256, generator=g).to(dev), 'track_query_boxes': (torch.rand(5, 4, generator=g) * 0.5 + 0.2).to(dev), 'image_id': torch.tensor([1], device=dev)}]
feats = []
with torch.no_grad():
    outs = []
    for _ in range(4):
        o, _, f, _, _ = model(img, [dict(target[0])], None)
        outs.append(o)
        feats.append(f[-1].tensors.clone())
print("split linears:", fused.split_linear_enabled())
for i in range(1, 4):
    print("backbone features run 0 vs %d: %.3g (max |f| %.3g)" % (i, float((feats[0] - feats[i]).abs().max()), float(feats[0].abs().max())))
for i in range(1, 4):
    print("model run 0 vs %d: logits %.3g boxes %.3g" % (i, float((outs[0]['pred_logits'] - outs[i]['pred_logits']).abs().max()), float((outs[0]['pred_boxes'] - outs[i]['pred_boxes']).abs().max())))
