"""Oracle: DANet act-time encoder forward, plain torch-CPU fp32 (TEST INFRASTRUCTURE ONLY).

Restates `DANet.get_latent_feature(x, "concate")` (reference
carla_perception/Networks/danet.py:216-238) functionally over a state_dict, for any input
size (the reference hard-codes the 5x8 layer-4 map: danet.py:91-92, intertask_att.py:17-18).
Pinned against the imported reference by tests/golden/make_golden.py (native 144x256
unmodified; 84x84 / 288x288 via module surgery on the imported instance).
"""
import numpy as np
import torch
import torch.nn.functional as F


def pre_process(rgb_u8, route_u8):
    """reference ppo_agent/agent.py:43-75 (use_vae branch).  rgb [S,H,W,3] u8, route [S,W,H] u8.
    Returns ([S,4,H,W] f32, mutated route u8) — the route normalisation is assigned back into
    the uint8 array (agent.py:51-54), truncating to {0,1} and mutating the caller's buffer."""
    rgb = np.array(rgb_u8 / 255., dtype=np.float32)          # float64 divide then cast (agent.py:46)
    img = rgb.transpose(0, 3, 1, 2)
    route = route_u8.copy()
    for i in range(route.shape[0]):
        mx = np.max(route[i]) * 1.0
        if mx > 0:
            route[i] = 1.0 * route[i] / mx                   # u8 store: truncation toward zero
    rf = np.array(route, dtype=np.float32).swapaxes(1, 2)
    rf = np.expand_dims(rf, 1)
    return np.concatenate([img, rf], axis=1), route


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def _bn(x, sd, pre):
    return F.batch_norm(x, _t(sd, pre + ".running_mean"), _t(sd, pre + ".running_var"),
                        _t(sd, pre + ".weight"), _t(sd, pre + ".bias"), False, 0.0, 1e-5)


def _conv(x, sd, pre, stride=1, pad=0):
    b = _t(sd, pre + ".bias") if (pre + ".bias") in sd else None
    return F.conv2d(x, _t(sd, pre + ".weight"), b, stride, pad)


def backbone(x, sd):
    """resnet.py:168-181 with BasicBlock :40-55."""
    x = F.relu(_bn(_conv(x, sd, "backbone.conv1", 2, 3), sd, "backbone.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            pre = "backbone.layer%d.%d" % (li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            idt = x
            o = F.relu(_bn(_conv(x, sd, pre + ".conv1", stride, 1), sd, pre + ".bn1"))
            o = _bn(_conv(o, sd, pre + ".conv2", 1, 1), sd, pre + ".bn2")
            if (pre + ".downsample.0.weight") in sd:
                idt = _bn(_conv(x, sd, pre + ".downsample.0", stride, 0), sd, pre + ".downsample.1")
            x = F.relu(o + idt)
    return x


def pam(x, sd, pre="da_head.sa"):
    """da_att.py:32-51."""
    b, C, h, w = x.shape
    q = _conv(x, sd, pre + ".query_conv").view(b, -1, h * w).permute(0, 2, 1)
    k = _conv(x, sd, pre + ".key_conv").view(b, -1, h * w)
    att = torch.softmax(torch.bmm(q, k), dim=-1)
    v = _conv(x, sd, pre + ".value_conv").view(b, -1, h * w)
    out = torch.bmm(v, att.permute(0, 2, 1)).view(b, C, h, w)
    return _t(sd, pre + ".gamma") * out + x


def cam(x, sd, pre="da_head.sc"):
    """da_att.py:63-83."""
    b, C, h, w = x.shape
    q = x.view(b, C, -1)
    e = torch.bmm(q, q.permute(0, 2, 1))
    e = torch.max(e, -1, keepdim=True)[0].expand_as(e) - e
    att = torch.softmax(e, dim=-1)
    out = torch.bmm(att, q).view(b, C, h, w)
    return _t(sd, pre + ".gamma") * out + x


def da_head(x, sd):
    """danet.py:43-69 (Dropout2d inert in eval)."""
    def cbr(x, nm):
        return F.relu(_bn(_conv(x, sd, "da_head.%s.0" % nm, 1, 1), sd, "da_head.%s.1" % nm))
    sa = cbr(pam(cbr(x, "conv5a"), sd), "conv51")
    sc = cbr(cam(cbr(x, "conv5c"), sd), "conv52")
    return _conv(sa + sc, sd, "da_head.conv8.1")


def inter_task_att(vis, bc, sd, z_dims=256):
    """intertask_att.py:121-176, att_type='transformer' (dropout inert in eval)."""
    b = vis.shape[0]
    vis = vis.reshape(b, -1)
    bc = bc.reshape(b, -1)

    def mlp(x, pre):
        y = F.linear(x, _t(sd, pre + ".1.weight"), _t(sd, pre + ".1.bias"))
        y = F.leaky_relu(y)
        return F.linear(y, _t(sd, pre + ".3.weight"), _t(sd, pre + ".3.bias"))
    P = "inter_task_att."
    vq, vk, vv = (mlp(vis, P + "visual_%s_layer" % r) for r in ("query", "key", "value"))
    bq, bk, bv = (mlp(bc, P + "bc_%s_layer" % r) for r in ("query", "key", "value"))
    temp = z_dims ** 0.5

    def cross(q, k, v):
        e = torch.bmm((q.view(b, 1, z_dims).permute(0, 2, 1)) / temp, k.view(b, 1, z_dims))
        att = torch.softmax(e, dim=-1)
        out = torch.bmm(v.view(b, 1, z_dims), att.permute(0, 2, 1)).view(b, -1)
        return out + v
    att_bc = cross(vq, bk, bv)
    att_vis = cross(bq, vk, vv)
    return att_vis, att_bc


def latent(x, sd, return_taps=False):
    """danet.py:216-238, mode 'concate'.  x [B,4,H,W] f32 -> [B,512]."""
    with torch.no_grad():
        x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        l4 = backbone(x, sd)
        da = da_head(l4, sd)
        vis = _conv(da, sd, "visual_conv")
        bc = _conv(da, sd, "bc_conv")
        av, ab = inter_task_att(vis, bc, sd)
        out = torch.cat((av, ab), dim=-1)
    if return_taps:
        return out, dict(layer4=l4, da=da, vis=vis, bc=bc)
    return out


def latent_feature(rgb_u8, route_u8, measurements, sd):
    """reference ppo_agent/agent.py:97-112: encoder latent ++ measurements.repeat(1,6) -> [S,530] f32."""
    x, _ = pre_process(rgb_u8, route_u8)
    lat = latent(torch.from_numpy(x), sd)
    m = torch.from_numpy(np.asarray(measurements)).repeat(1, 6)
    return torch.cat([lat, m], dim=-1).float()
