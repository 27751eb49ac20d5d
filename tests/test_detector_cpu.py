"""CPU tier: the caller-side (pure torch) pieces of jmodt_amd/detector.py — BatchNorm folding, the attention
fusion as accumulating GEMMs, the composed deconvolution + fusion convolution, the Conv1d heads, the RCNN
xyz lift — against the un-fused chained oracle (oracle/pipeline.py) and the parameter containers' own
module forwards.  The jm_* operators in between need a GPU (tests/test_gpu_detector.py)."""
import numpy as np
import pytest
import torch

from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
from oracle.pipeline import Chain


@pytest.fixture(scope="module")
def eng():
    torch.manual_seed(0)
    e = DetectAffinityEngine(DetectorConfig.tiny())
    g = torch.Generator().manual_seed(1)
    for m in e.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return e.double()


def test_full_config_matches_reference_parameter_count():
    # 66.9 MB of fp32 parameters in the joint model, 525,825 per affinity head (§8 a14)
    e = DetectAffinityEngine()
    assert sum(p.numel() for p in e.parameters()) == 16_732_011
    assert sum(p.numel() for p in e.rcnn_net.link_layer.parameters()) == 525_825
    assert e.cfg.rpn_reg_channels == 76 and e.cfg.rcnn_reg_channels == 46
    keys = e.state_dict().keys()
    for k in ("rpn.backbone_net.SA_modules.3.mlps.1.layer2.bn.bn.running_var", "rpn.backbone_net.Img_Block.0.conv1.weight",
              "rpn.backbone_net.Fusion_Conv.2.IA_Layer.fc3.bias", "rpn.backbone_net.DeConv.3.weight",
              "rpn.backbone_net.final_fusion_img_point.IA_Layer.conv1.1.running_mean",
              "rpn.backbone_net.FP_modules.0.mlp.layer1.conv.weight", "rpn.rpn_reg_layer.2.conv.weight",
              "rcnn_net.xyz_up_layer.layer1.conv.bias", "rcnn_net.merge_down_layer.layer0.conv.weight",
              "rcnn_net.SA_modules.2.mlps.0.layer2.conv.weight", "rcnn_net.cls_layer.3.conv.weight",
              "rcnn_net.link_layer.3.conv.bias", "rcnn_net.se_layer.0.conv.weight"):
        assert k in keys, k


def test_attention_fusion_folded_matches_module_and_chain(eng):
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    net = eng.rpn.backbone_net
    g = torch.Generator().manual_seed(2)
    for i, mod in enumerate(net.Fusion_Conv):
        ic, pc = mod.IA_Layer.fc1.in_features, mod.IA_Layer.fc2.in_features
        P = torch.randn(2, pc, 37, generator=g).double()
        I = torch.randn(2, ic, 37, generator=g).double()
        got = eng._attention_fusion(f"t{i}", mod, P, I)
        assert torch.allclose(got, mod(P, I), atol=1e-10)
        assert torch.allclose(got, chain._attention_fusion(f"rpn.backbone_net.Fusion_Conv.{i}", P, I), atol=1e-10)


def test_image_fusion_map_composition_matches_literal(eng):
    import torch.nn.functional as F
    net, cfg = eng.rpn.backbone_net, eng.cfg
    g = torch.Generator().manual_seed(3)
    H, W = 32, 64
    maps = [torch.randn(2, c, H >> (i + 1), W >> (i + 1), generator=g).double() for i, c in enumerate(cfg.img_channels[1:])]
    got = eng._image_fusion_map([m.contiguous(memory_format=torch.channels_last) for m in maps])
    cat = torch.cat([net.DeConv[i](m) for i, m in enumerate(maps)], dim=1)
    want = F.relu(net.image_fusion_bn(net.image_fusion_conv(cat)))
    assert got.shape == want.shape == (2, cfg.img_features_channel // 4, H, W)
    assert torch.allclose(got, want, atol=1e-10)


def test_heads_and_image_blocks_match_chain(eng):
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, eng.cfg.fp_mlps[0][-1], 50, generator=g).double()
    assert torch.allclose(eng._head_forward("a", eng.rpn.rpn_cls_layer, x), chain._head(x, "rpn.rpn_cls_layer"), atol=1e-10)
    assert torch.allclose(eng._head_forward("b", eng.rpn.rpn_reg_layer, x), chain._head(x, "rpn.rpn_reg_layer"), atol=1e-10)
    f = torch.randn(5, eng.cfg.rcnn_sa_mlps[-1][-1], 1, generator=g).double()
    assert torch.allclose(eng._head_forward("c", eng.rcnn_net.reg_layer, f), chain._head(f, "rcnn_net.reg_layer"), atol=1e-10)
    assert torch.allclose(eng.rcnn_net.link_layer(f), chain._head(f, "rcnn_net.link_layer"), atol=1e-10)
    img = torch.randn(1, 3, 16, 32, generator=g).double()
    blk = eng.rpn.backbone_net.Img_Block[0]
    sd = chain.sd
    import torch.nn.functional as F
    y = F.conv2d(img, sd["rpn.backbone_net.Img_Block.0.conv1.weight"], None, 1, 1)
    y = torch.relu(chain._bn(y, "rpn.backbone_net.Img_Block.0.bn1"))
    y = F.conv2d(y, sd["rpn.backbone_net.Img_Block.0.conv2.weight"], None, 2, 1)
    assert torch.allclose(blk(img), y, atol=1e-10)


def test_rcnn_lift_matches_chain(eng, monkeypatch):
    """xyz_up_layer + merge_down_layer on strided row views == the reference's transpose / cat / SharedMLP form"""
    chain = Chain(eng.state_dict(), eng.cfg, torch.float64)
    cfg = eng.cfg
    g = torch.Generator().manual_seed(5)
    R, S, C = 6, cfg.rcnn_num_points, cfg.fp_mlps[0][-1]
    pts = torch.randn(R, S, 5 + C, generator=g).double()
    captured = {}

    class Stop(Exception):
        pass

    def fake_sa(xyz, feats, *a, **k):
        captured["feats"] = feats
        raise Stop
    monkeypatch.setattr(eng.rcnn_net.SA_modules[0], "forward", fake_sa)
    with pytest.raises(Stop):
        eng.rcnn_forward(pts)
    xyz_in = pts[..., 0:5].transpose(1, 2).contiguous().unsqueeze(3)
    rpn_feat = pts[..., 5:].transpose(1, 2).contiguous().unsqueeze(3)
    merged = chain._shared_mlp(torch.cat((chain._shared_mlp(xyz_in, "rcnn_net.xyz_up_layer"), rpn_feat), 1),
                               "rcnn_net.merge_down_layer").squeeze(3)
    assert captured["feats"].shape == merged.shape and torch.allclose(captured["feats"], merged, atol=1e-10)
