"""Host-side pieces of the training step against fixtures produced by the reference's own Python."""
import os

import numpy as np
import torch

from neuralgaussiansplatting_amd import loss_utils, train_step

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_position_lr_schedule_matches_reference_fixture():
    z = np.load(os.path.join(GOLDEN, "expon_lr.npz"))
    for i, s in enumerate(z["steps"]):
        a = train_step.expon_lr(int(s), 0.00016, 0.0000016, lr_delay_mult=0.01, max_steps=30_000)
        b = train_step.expon_lr(int(s), 0.01, 0.0001, lr_delay_steps=500, lr_delay_mult=0.05, max_steps=5000)
        assert abs(a - z["default"][i]) <= 1e-15 * max(1.0, abs(z["default"][i]) * 1e3)
        assert abs(b - z["delayed"][i]) <= 1e-12
    assert train_step.expon_lr(-1, 1e-3, 1e-4) == 0.0 and train_step.expon_lr(5, 0.0, 0.0) == 0.0
    o = train_step.OptimizationParams()
    assert (o.feature_lr, o.opacity_lr, o.scaling_lr, o.rotation_lr, o.lambda_dssim) == (0.0025, 0.05, 0.005, 0.001, 0.2)


def test_torch_training_loss_and_gradient_match_reference_fixture():
    z = np.load(os.path.join(GOLDEN, "train_loss.npz"))
    img = torch.tensor(z["img"], requires_grad=True)
    gt = torch.tensor(z["gt"])
    loss = loss_utils.training_loss(img, gt, float(z["lambda_dssim"]))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    np.testing.assert_allclose(img.grad.numpy(), z["grad"], rtol=0, atol=1e-9)


def test_inverse_sigmoid_round_trip():
    x = torch.tensor([0.1, 0.5, 0.95])
    assert torch.allclose(torch.sigmoid(train_step.inverse_sigmoid(x)), x, atol=1e-6)
