"""CPU: the restated BatchSemiHardTripletLoss (oracle/carel_oracle_st.py, vectorised as the sentence-transformers package
publishes it) against an independent loop-form statement of its definition -- parity with the package itself is UNPINNED
(absent from this container and from /root/reference; the reference holds no outputs for this path)."""
import torch

from oracle import carel_oracle_st as ST


def test_vectorised_restatement_equals_the_definition():
    g = torch.Generator().manual_seed(0)
    for trial in range(12):
        B, H = [16, 7, 2, 33][trial % 4], 24
        emb = torch.randn((B, H), generator=g) * (0.3 + trial)
        labels = torch.randint(0, [3, 7, 1, 2][trial % 4], (B,), generator=g)
        if trial % 4 == 2:
            labels = torch.zeros(B, dtype=torch.long)              # the English script as committed: every sentence label 0
        for margin in (0.5, 4.45, 400.0):
            a = ST.batch_semi_hard_triplet_loss(labels, emb, margin)
            b = ST.triplet_loss_by_definition(labels, emb, margin)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (trial, margin, float(a), float(b))


def test_mean_pooling_ignores_padding_and_duplicates_have_zero_distance():
    h = torch.arange(2 * 4 * 3, dtype=torch.float32).reshape(2, 4, 3)
    m = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1]])
    p = ST.mean_pool(h, m)
    assert torch.allclose(p[0], h[0, :2].mean(0)) and torch.allclose(p[1], h[1].mean(0))
    e = torch.tensor([[1.0, 2.0], [1.0, 2.0], [4.0, 6.0]], requires_grad=True)
    d = ST.euclidean_distance(e)
    assert float(d[0, 1]) == 0.0 and abs(float(d[0, 2]) - 5.0) < 1e-6
    d.sum().backward()
    assert torch.isfinite(e.grad).all()
