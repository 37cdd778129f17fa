"""Pairwise instance distances on the GPU + host assignment against the
torch.linalg.norm / scipy results recorded in tests/golden/match_small.npz."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pairwise_l2_matches_reference_cost(matchfx, device):
    from mass_amd.utils.experimentation import pairwise_distance
    tags = sorted({k[:-3] for k in matchfx.files if k.endswith("_f0")})
    for t in tags:
        f0, f1 = torch.tensor(matchfx[t + "_f0"]).to(device), torch.tensor(matchfx[t + "_f1"]).to(device)
        want = matchfx[t + "_cost"]
        got = pairwise_distance(f0, f1, metric="l2").cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=t)      # tolerance: SURVEY 8(d) config 4
        gemm = pairwise_distance(f0, f1, metric="l2_gemm").cpu().numpy()
        # |a|^2+|b|^2-2ab cancels when a ~ b: absolute error ~ 1e-6 * |a|^2 on d^2
        scale = float((f0 ** 2).sum(1).max() + (f1 ** 2).sum(1).max())
        assert np.abs(gemm ** 2 - want.astype(np.float64) ** 2).max() <= 4e-6 * scale, t


def test_match_instances_equals_scipy_assignment(matchfx, device):
    from mass_amd.utils.experimentation import match_instances
    tags = sorted({k[:-3] for k in matchfx.files if k.endswith("_f0")})
    for t in tags:
        f0, f1 = torch.tensor(matchfx[t + "_f0"]).to(device), torch.tensor(matchfx[t + "_f1"]).to(device)
        cost, rows, cols = match_instances(f0, f1)
        assert np.array_equal(rows, matchfx[t + "_rows"]) and np.array_equal(cols, matchfx[t + "_cols"]), t


def test_config4_200x200x1024(matchfx, device):
    """BASELINE config 4: N0 = N1 = 200, D = 1024, seeds 0 / 1."""
    from mass_amd.utils.experimentation import pairwise_distance, match_instances
    f0 = torch.randn(200, 1024, generator=torch.Generator().manual_seed(0))
    f1 = torch.randn(200, 1024, generator=torch.Generator().manual_seed(1))
    if hashlib.sha256(f0.numpy().tobytes() + f1.numpy().tobytes()).hexdigest() != str(matchfx["cfg4_input_sha256"]):
        pytest.skip("torch RNG stream differs from the recorded one")
    f0, f1 = f0.to(device), f1.to(device)
    for metric in ("l2", "l2_gemm"):
        got = pairwise_distance(f0, f1, metric=metric).cpu().numpy()
        np.testing.assert_allclose(got, matchfx["cfg4_cost"], rtol=1e-5)
    cost, rows, cols = match_instances(f0, f1)
    assert np.array_equal(rows, matchfx["cfg4_rows"]) and np.array_equal(cols, matchfx["cfg4_cols"])
    cost_g, rows_g, cols_g = match_instances(f0, f1, metric="l2_gemm")
    assert np.array_equal(cols_g, matchfx["cfg4_cols"])
    # cosine (not in the reference) against torch
    cos = pairwise_distance(f0, f1, metric="cosine").cpu()
    want = 1 - torch.nn.functional.cosine_similarity(f0.cpu()[:, None], f1.cpu()[None], dim=2)
    np.testing.assert_allclose(cos.numpy(), want.numpy(), rtol=1e-4, atol=2e-6)


def test_goal_distance_and_ragged_shapes(device):
    from mass_amd.utils.experimentation import pairwise_distance
    g = torch.Generator().manual_seed(2)
    for n0, n1, d in ((1, 1, 1), (7, 3, 3), (65, 130, 17), (33, 64, 100)):
        a, b = torch.randn(n0, d, generator=g), torch.randn(n1, d, generator=g)
        want = torch.linalg.norm(a.unsqueeze(1) - b.unsqueeze(0), dim=2).numpy()
        for metric in ("l2", "l2_gemm"):
            got = pairwise_distance(a.to(device), b.to(device), metric=metric).cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5, err_msg=f"{metric} {n0}x{n1}x{d}")
    assert pairwise_distance(torch.zeros(0, 4, device=device), torch.zeros(3, 4, device=device)).shape == (0, 3)
