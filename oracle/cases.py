"""Seeded case construction shared by the golden generator (which applies it to the reference's
class) and the tests (which apply it to the drop-in / the restatements): same RNG calls in the
same order, so both sides hold bit-identical weights and inputs.  TEST INFRASTRUCTURE ONLY."""
import torch


def bio_case(cls, seed, n, cin, ncls, h, w):
    """cls(cin, ncls) under manual_seed(seed), default torch init, then every 1-D parameter (BN
    affine, conv / deconv biases) perturbed so none of them is at its trivial initial value."""
    torch.manual_seed(seed)
    m = cls(cin, ncls).train()
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    x = torch.randn(n, cin, h, w, generator=g)
    t = torch.randint(0, ncls, (n, h, w), generator=g)
    return m, x, t


def bio_weights_match(z, state_dict):
    """True when `state_dict` (rebuilt from the fixture's seed) carries the checksums the generator
    recorded from the reference's own module."""
    import numpy as np
    assert list(state_dict.keys()) == [str(k) for k in z["keys"]]
    for k, v in state_dict.items():
        v = v.double()
        got = np.array([float(v.sum()), float(v.abs().sum())])
        if not np.allclose(got, z["wsum/" + k], rtol=1e-12, atol=1e-12):
            return False
    return True


def bio_grad_errors(z, grads, rel):
    """Compare a {name: ndarray} gradient set with the fixture (full tensors, or L2 norm + sum +
    strided sample for the large ones).  Returns a list of failure strings."""
    import numpy as np
    bad = []
    for key in z.files:
        kind, _, name = key.partition("/")
        if kind not in ("g", "gs", "gn"):
            continue
        g = np.asarray(grads[name], np.float64)
        if name.endswith((".0.bias", ".3.bias")):   # conv bias in front of BN: zero up to rounding on both sides
            if kind != "gn" and (np.abs(g).max() > 1e-6 or np.abs(z[key]).max() > 1e-6):
                bad.append(f"{name}: conv-bias gradient not ~0")
            continue
        if kind == "g":
            ref, got = z[key], g
        elif kind == "gs":
            ref, got = z[key], g.reshape(-1)[::211]
        else:
            ref, got = z[key][:1], np.array([np.sqrt((g ** 2).sum())])
        tol = rel * max(float(np.abs(ref).max()), 1e-4)
        err = float(np.abs(got - ref).max())
        if err > tol:
            bad.append(f"{key}: max err {err:.3e} > {tol:.3e}")
    return bad


def ynet_case(cls, seed, in_ch, n_cls, feat, shape):
    """cls(in_ch, n_cls, init_features=feat) under manual_seed(seed) with the BatchNorm affine parameters moved off
    their trivial initial values -- the recipe of tools/gen_golden.py::unet_case, shared so that a network too large
    to store (UNet(1,8,32): 31 MB of weights) is rebuilt bit-identically on both sides."""
    torch.manual_seed(seed)
    m = cls(in_ch, n_cls, init_features=feat)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    B, H, W = shape
    x = torch.randn(B, in_ch, H, W, generator=g)
    t = torch.randint(0, n_cls, (B, H, W), generator=g)
    return m.train(), x, t


def grad_summary(g):
    """[L2 norm, sum, sum of |.|] + a strided sample: what a fixture keeps of a gradient too large to store"""
    import numpy as np
    g = np.asarray(g, np.float64).reshape(-1)
    return np.array([np.sqrt((g ** 2).sum()), g.sum(), np.abs(g).sum()]), g[::97].copy()
