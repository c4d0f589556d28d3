"""RBF-MMD operator (MMDStatistic/pdist, ref :537-596) through the C ABI vs. golden vectors + oracle."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L
from oracle import carel_oracle as O

pytestmark = pytest.mark.gpu


def run_mmd(s1, s2, alphas, ret_matrix=False, grad=None):
    n1, n2, d = s1.shape[0], s2.shape[0], s1.shape[1]
    a = L.MmdArgs()
    a.s1, a.s2, a.ld1, a.ld2 = s1.data_ptr(), s2.data_ptr(), s1.stride(0), s2.stride(0)
    a.n1, a.n2, a.d, a.n_alphas, a.eps = n1, n2, d, len(alphas), 1e-5
    for i, v in enumerate(alphas):
        a.alphas[i] = v
    out = torch.zeros(1, device="cuda")
    kern = torch.zeros((n1 + n2, n1 + n2), device="cuda") if ret_matrix else None
    a.mmd_out = out.data_ptr()
    a.kernels_out = None if kern is None else kern.data_ptr()
    lib = L.load()
    L.check(lib.carel_rbf_mmd_fwd(C.byref(a), L.current_stream()), "mmd fwd")
    g1 = g2 = None
    if grad is not None:
        g = torch.tensor([grad], device="cuda", dtype=torch.float32)
        g1, g2 = torch.zeros((n1, d), device="cuda"), torch.zeros((n2, d), device="cuda")
        a.grad_mmd, a.g1, a.g2 = g.data_ptr(), g1.data_ptr(), g2.data_ptr()
        L.check(lib.carel_rbf_mmd_bwd(C.byref(a), L.current_stream()), "mmd bwd")
    torch.cuda.synchronize()
    return out.item(), kern, g1, g2


def test_mmd_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "statistics.npz"), allow_pickle=False)
    for tag in "abcde":
        s1, s2 = torch.from_numpy(z[f"{tag}_s1"]).cuda(), torch.from_numpy(z[f"{tag}_s2"]).cuda()
        mmd, kern, g1, g2 = run_mmd(s1, s2, [0.1], ret_matrix=True, grad=-1.0)
        np.testing.assert_allclose(mmd, float(z[f"{tag}_mmd"]), rtol=2e-5, atol=2e-7)
        _, kref = O.mmd_statistic(s1.cpu(), s2.cpu(), [0.1], ret_matrix=True)
        np.testing.assert_allclose(kern.cpu().numpy(), kref.numpy(), rtol=1e-5, atol=1e-6)
        # gradient of -mmd (what the training loss back-propagates, ref :233)
        scale = np.abs(z[f"{tag}_g1"]).max()
        np.testing.assert_allclose(g1.cpu().numpy(), z[f"{tag}_g1"], rtol=2e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(g2.cpu().numpy(), z[f"{tag}_g2"], rtol=2e-4, atol=2e-5 * scale)
        mmd3, _, _, _ = run_mmd(s1, s2, [0.1, 0.5, 2.0])
        np.testing.assert_allclose(mmd3, float(z[f"{tag}_mmd3"]), rtol=2e-5, atol=2e-7)


def test_mmd_properties_and_errors():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((64, 24), generator=g).cuda()
    # identical samples: cross and within kernels agree up to the diagonal treatment -> small value
    m_same, _, _, _ = run_mmd(x, x.clone(), [0.1])
    y = (torch.randn((64, 24), generator=g) * 3 + 2).cuda()
    m_far, _, _, _ = run_mmd(x, y, [0.1])
    assert m_far > m_same
    # strided views (columns of a wider matrix) are accepted via ld
    wide = torch.randn((64, 48), generator=g).cuda()
    m1, _, _, _ = run_mmd(wide[:, :24], wide[:, 24:], [0.1])
    m2, _, _, _ = run_mmd(wide[:, :24].contiguous(), wide[:, 24:].contiguous(), [0.1])
    assert m1 == m2
    with pytest.raises(L.CarelError):
        run_mmd(x[:1], x[:1], [0.1])      # n = 1 divides by zero in the reference


def test_hsic_golden_and_gradients(golden_dir):
    """HSIC ablation head (drl_classifier_ec_hsic.py:529-547) vs the reference's values and the oracle's autograd."""
    from carel_vae_amd import HSIC
    z = np.load(os.path.join(golden_dir, "statistics.npz"), allow_pickle=False)
    for tag in "ab":
        x, y = torch.from_numpy(z[f"h{tag}_x"]), torch.from_numpy(z[f"h{tag}_y"])
        xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
        h = HSIC(xg, yg)
        np.testing.assert_allclose(h.item(), float(z[f"h{tag}_hsic"]), rtol=2e-4, atol=1e-7)
        (3.0 * h).backward()
        xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        (3.0 * O.hsic_statistic(xo, yo)).backward()
        sc = float(xo.grad.abs().max())
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), rtol=2e-3, atol=3e-3 * sc)   # gradients are ~1e-7: fp32 noise of both sides
        np.testing.assert_allclose(yg.grad.cpu().numpy(), yo.grad.numpy(), rtol=2e-3, atol=3e-3 * sc)
    with pytest.raises(L.CarelError):
        HSIC(torch.zeros(1, 24, device="cuda"), torch.zeros(1, 24, device="cuda"))


def _gslice(m, k):
    from tests.test_oracle_golden import gslice
    return gslice(m, k)


def test_exported_statistics_wrappers_against_the_reference_fixture(golden_dir):
    """The drop-in Python surface itself -- carel_vae_amd.MMDStatistic(n1, n2)(...), .pdist, .HSIC with autograd -- against
    the values and gradients the reference's own classes produced (tests/golden/statistics.npz, ref :537-596 and
    drl_classifier_ec_hsic.py:540-547)."""
    import carel_vae_amd as M
    z = np.load(os.path.join(golden_dir, "statistics.npz"), allow_pickle=False)
    for tag in "abcde":
        s1 = torch.from_numpy(z[f"{tag}_s1"]).cuda().requires_grad_(True)
        s2 = torch.from_numpy(z[f"{tag}_s2"]).cuda().requires_grad_(True)
        stat = M.MMDStatistic(s1.shape[0], s2.shape[0])
        mmd, kern = stat(s1, s2, [0.1], ret_matrix=True)
        np.testing.assert_allclose(mmd.item(), float(z[f"{tag}_mmd"]), rtol=2e-5, atol=2e-7)
        np.testing.assert_allclose(_gslice(kern.detach().cpu(), 32), z[f"{tag}_kern_slice"], rtol=1e-5, atol=1e-6)
        (-mmd).backward()
        scale = np.abs(z[f"{tag}_g1"]).max()
        np.testing.assert_allclose(s1.grad.cpu().numpy(), z[f"{tag}_g1"], rtol=2e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(s2.grad.cpu().numpy(), z[f"{tag}_g2"], rtol=2e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(stat(s1.detach(), s2.detach(), [0.1, 0.5, 2.0]).item(), float(z[f"{tag}_mmd3"]), rtol=2e-5, atol=2e-7)
        dist = M.pdist(s1.detach(), s2.detach())
        np.testing.assert_allclose(_gslice(dist.cpu(), 32), z[f"{tag}_pdist_slice"], rtol=2e-6, atol=1e-6)
    for tag in "ab":
        x, y = torch.from_numpy(z[f"h{tag}_x"]).cuda(), torch.from_numpy(z[f"h{tag}_y"]).cuda()
        np.testing.assert_allclose(M.HSIC(x, y).item(), float(z[f"h{tag}_hsic"]), rtol=2e-4, atol=1e-7)


def test_pdist_far_near_and_gradients():
    """pdist must not go through exp/log: samples 30 apart per coordinate (d2 ~ 20 000, exp underflows) and samples that
    coincide (d2 = 0 -> sqrt(eps)); gradients against torch autograd of the reference's formula."""
    import carel_vae_amd as M
    g = torch.Generator().manual_seed(3)
    a = torch.randn((40, 24), generator=g)
    b = torch.randn((56, 24), generator=g) + 30.0
    b[0] = a[0]                                                   # a coincident pair
    def ref(x, y, eps=1e-5):
        n = (x ** 2).sum(1, keepdim=True) + (y ** 2).sum(1, keepdim=True).t()
        return torch.sqrt(eps + torch.abs(n - 2 * x.mm(y.t())))
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    want = ref(ad, bd)
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    got = M.pdist(ag, bg)
    assert torch.isfinite(got).all() and got.max().item() > 100.0
    far = want > 1.0
    np.testing.assert_allclose(got.detach().cpu().double()[far].numpy(), want.detach()[far].numpy(), rtol=3e-5)
    # d2 is a difference of ~1e4-sized fp32 terms for the far pairs; near pairs: |d2| error ~1e-5 absolute
    assert abs(got[0, 0].item() - math.sqrt(1e-5)) < 2e-3
    w = torch.randn(want.shape, generator=g).double()
    w[0, 0] = 0.0                                                # the coincident pair's gradient is sign-undefined
    (want * w).sum().backward()
    (got * w.float().cuda()).sum().backward()
    np.testing.assert_allclose(ag.grad.cpu().numpy(), ad.grad.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), bd.grad.numpy(), rtol=2e-3, atol=2e-3)
    with pytest.raises(NotImplementedError):
        M.pdist(ag, bg, norm=1)
