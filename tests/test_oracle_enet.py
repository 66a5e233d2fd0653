"""The oracle's EnhanceNet-generator restatement (forward and backward, enet/enet/model_enet.py:8-115, :331-337)
against an independent implementation: torch CPU float64 convolutions with autograd.  (TensorFlow is not
installed, so the reference itself cannot run: this pins the restatement's calculus, not TF's kernels.)"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import oracle as O

LAYERS = [(3, 3, 64)] + [(3, 64, 64), (1, 64, 64)] * 10 + [(3, 64, 64)] * 3 + [(3, 64, 3)]


def test_generator_forward_backward_vs_autograd():
    rng = np.random.default_rng(5)
    params = [(rng.normal(0, 1 / np.sqrt(k * k * ci), (k, k, ci, co)), rng.uniform(-.1, .1, co)) for k, ci, co in LAYERS]
    sd = rng.uniform(-1, 1, (2, 5, 4, 3))
    bq = rng.uniform(-1, 1, (2, 20, 16, 3))
    d_sr = rng.normal(0, 1, (2, 20, 16, 3))
    sr, ins = O.enet_generator_forward(sd, bq, params, keep=True)
    grads = O.enet_generator_backward(ins, d_sr, params)

    tp = [(torch.tensor(k, requires_grad=True), torch.tensor(b, requires_grad=True)) for k, b in params]

    def conv(x, k, b):
        return F.conv2d(x, k.permute(3, 2, 0, 1), b, padding=k.shape[0] // 2)

    t = F.relu(conv(torch.tensor(sd).permute(0, 3, 1, 2), *tp[0]))
    i = 1
    for _ in range(10):
        y = F.relu(conv(t, *tp[i]))
        t = F.relu(t + conv(y, *tp[i + 1]))
        i += 2
    for _ in range(2):
        t = F.relu(conv(F.interpolate(t, scale_factor=2, mode='nearest'), *tp[i]))
        i += 1
    t = F.relu(conv(t, *tp[i]))
    out = conv(t, *tp[i + 1]) + torch.tensor(bq).permute(0, 3, 1, 2)
    np.testing.assert_allclose(out.permute(0, 2, 3, 1).detach().numpy(), sr, rtol=0, atol=1e-12)
    (out * torch.tensor(d_sr).permute(0, 3, 1, 2)).sum().backward()
    for (k, b), (dw, db) in zip(tp, grads):
        np.testing.assert_allclose(k.grad.numpy(), dw, rtol=1e-10, atol=1e-12 * np.abs(dw).max())
        np.testing.assert_allclose(b.grad.numpy(), db, rtol=1e-10, atol=1e-12 * np.abs(db).max())
