"""Seeded synthetic weights / observations for parity tests and bench.py.

There is no network and the pretrained encoder is a Google-Drive download (reference
README.md:19), so every fixture, test and benchmark uses weights regenerated from seeds.
Generation is keyed by tensor *name* (crc32(name) ^ seed) so any subset can be rebuilt
independently and the golden generator, the tests and the bench all see identical arrays.

Encoder key names / shapes follow the reference state_dict
(carla_perception/Networks/danet.py:72-110, danet_blocks/resnet.py:103-166,
danet_blocks/da_att.py:24-29,58-61, danet_blocks/intertask_att.py:39-80) restricted to the
tensors `get_latent_feature` actually reads.
"""
import zlib

import numpy as np

EPS_BN = 1e-5


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def encoder_spec(feat_h, feat_w, in_ch=4):
    """[(key, shape, kind)] for the act-time encoder at a layer-4 map of feat_h x feat_w."""
    spec = []

    def conv(name, co, ci, k, bias):
        spec.append((name + ".weight", (co, ci, k, k), "conv"))
        if bias:
            spec.append((name + ".bias", (co,), "bias"))

    def bn(name, c, wkind="bn_w"):
        spec.append((name + ".weight", (c,), wkind))
        spec.append((name + ".bias", (c,), "bias"))
        spec.append((name + ".running_mean", (c,), "bias"))
        spec.append((name + ".running_var", (c,), "bn_w"))

    conv("backbone.conv1", 64, in_ch, 7, True)          # resnet.py:111-112 (bias_first=True)
    bn("backbone.bn1", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            pre = "backbone.layer%d.%d" % (li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            conv(pre + ".conv1", planes, inpl, 3, False)
            bn(pre + ".bn1", planes)
            conv(pre + ".conv2", planes, planes, 3, False)
            bn(pre + ".bn2", planes)
            if stride != 1 or inpl != planes:
                conv(pre + ".downsample.0", planes, inpl, 1, False)
                bn(pre + ".downsample.1", planes)
            inpl = planes
    for nm, ci in (("conv5a", 512), ("conv5c", 512), ("conv51", 128), ("conv52", 128)):
        conv("da_head.%s.0" % nm, 128, ci, 3, False)
        # small BN gain in front of PAM/CAM keeps the attention energies O(1..10) so the
        # softmaxes are neither one-hot nor uniform (a trained net's statistics do the same)
        bn("da_head.%s.1" % nm, 128, "bn_w_att" if nm in ("conv5a", "conv5c") else "bn_w")
    spec.append(("da_head.sa.gamma", (1,), "gamma"))
    conv("da_head.sa.query_conv", 16, 128, 1, True)
    conv("da_head.sa.key_conv", 16, 128, 1, True)
    conv("da_head.sa.value_conv", 128, 128, 1, True)
    spec.append(("da_head.sc.gamma", (1,), "gamma"))
    conv("da_head.conv8.1", 512, 128, 1, True)
    conv("visual_conv", 512, 512, 1, True)
    conv("bc_conv", 512, 512, 1, True)
    in_dim = 512 * feat_h * feat_w
    for br in ("visual", "bc"):
        for role in ("query", "key", "value"):
            pre = "inter_task_att.%s_%s_layer" % (br, role)
            spec.append((pre + ".1.weight", (512, in_dim), "lin"))
            spec.append((pre + ".1.bias", (512,), "bias"))
            spec.append((pre + ".3.weight", (256, 512), "lin"))
            spec.append((pre + ".3.bias", (256,), "bias"))
    return spec


def make_tensor(name, shape, kind, seed):
    r = _rs(name, seed)
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        # slightly below He-init so 8 residual blocks keep O(1) activations
        return (r.standard_normal(shape) * np.sqrt(1.6 / fan_in)).astype(np.float32)
    if kind == "lin":
        return (r.standard_normal(shape) * np.sqrt(1.0 / shape[1])).astype(np.float32)
    if kind == "bias":
        return (r.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == "bn_w":
        return r.uniform(0.6, 1.4, shape).astype(np.float32)
    if kind == "bn_w_att":
        return r.uniform(0.1, 0.3, shape).astype(np.float32)
    if kind == "gamma":
        return np.full(shape, 0.5, np.float32)      # reference inits 0 (da_att.py:29,61): would hide PAM/CAM
    raise ValueError(kind)


def encoder_state(feat_h, feat_w, seed=0):
    """dict name -> float32 ndarray (reference state_dict layout, OIHW conv weights)."""
    return {k: make_tensor(k, s, kd, seed) for k, s, kd in encoder_spec(feat_h, feat_w)}


def feat_hw(H, W):
    """layer-4 map size of the ResNet-18 trunk for an HxW input (resnet.py:111-115,152-166)."""
    def down(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = down(H, 7, 2, 3), down(W, 7, 2, 3)
    h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
    for _ in range(3):
        h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
    return h, w


# ----------------------------------------------------------------------------- PPO nets
HEADS = ("steer", "throttle")


def ppo_net_spec(obs_dim=530, n_out=None, command_num=4):
    """[(model_name, param_name, shape, kind)] in reference model_dict naming
    (ppo_agent/models.py:100-125, distributions.py:34-40)."""
    n_out = n_out or {"steer": 33, "throttle": 3}
    spec = []
    for c in range(command_num):
        for hd in HEADS:
            m = "%s_ppo_%d" % (hd, c)
            for tower, last in (("control.linear", n_out[hd]), ("critic", 1)):
                spec.append((m, tower + ".0.weight", (128, obs_dim), "lin"))
                spec.append((m, tower + ".0.bias", (128,), "bias"))
                spec.append((m, tower + ".2.weight", (128, 128), "lin"))
                spec.append((m, tower + ".2.bias", (128,), "bias"))
                spec.append((m, tower + ".4.weight", (last, 128), "lin"))
                spec.append((m, tower + ".4.bias", (last,), "bias"))
            m = "%s_lstm_%d" % (hd, c)
            spec.append((m, "rnn.weight_ih", (4 * obs_dim, obs_dim), "lin"))
            spec.append((m, "rnn.weight_hh", (4 * obs_dim, obs_dim), "lin"))
            spec.append((m, "rnn.bias_ih", (4 * obs_dim,), "bias"))
            spec.append((m, "rnn.bias_hh", (4 * obs_dim,), "bias"))
    return spec


def ppo_state(seed=0, obs_dim=530, n_out=None, command_num=4):
    """{model_name: {param_name: ndarray}} — random (not orthogonal) but well-scaled, non-zero
    biases so every gradient path is exercised."""
    out = {}
    for m, p, shape, kind in ppo_net_spec(obs_dim, n_out, command_num):
        out.setdefault(m, {})[p] = make_tensor(m + "." + p, shape, kind, seed)
    return out


# ----------------------------------------------------------------------------- observations
def synth_rollout(T, H, W, seq=8, seed=1234, done_p=0.02):
    """Synthetic rollout of SURVEY.md §8(d): sliding 8-frame window (env_wrapper.py:899-904),
    u8 RGB [S,H,W,3], u8 route mask [S,W,H] (stored transposed like the reference), f64
    measurements [S,3], command, per-head reward and done flags."""
    rng = np.random.RandomState(seed)
    rgb = rng.randint(0, 256, (seq, H, W, 3)).astype(np.uint8)
    route = ((rng.rand(seq, W, H) < 0.15) * 255).astype(np.uint8)
    meas = rng.rand(seq, 3)
    steps = []
    for t in range(T):
        if t > 0:
            rgb = np.concatenate([rgb[1:], rng.randint(0, 256, (1, H, W, 3)).astype(np.uint8)], 0)
            route = np.concatenate([route[1:], ((rng.rand(1, W, H) < 0.15) * 255).astype(np.uint8)], 0)
            meas = np.concatenate([meas[1:], rng.rand(1, 3)], 0)
        steps.append(dict(rgb=rgb, route_fig=route.copy(), measurements=meas,
                          command=int(rng.randint(0, 4)),
                          reward=rng.rand(2).astype(np.float32),
                          done=(rng.rand(2) < done_p)))
    return steps
