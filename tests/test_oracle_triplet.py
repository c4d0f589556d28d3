"""CPU: the restated BatchSemiHardTripletLoss (oracle/carel_oracle_st.py, vectorised as the sentence-transformers package
publishes it) against an independent loop-form statement of its definition -- parity with the package itself is UNPINNED
(absent from this container and from /root/reference; the reference holds no outputs for this path)."""
import pytest
import torch

from oracle import carel_oracle as O
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


def test_mpnet_encoder_restatement_against_installed_transformers():
    """The English script loads all-mpnet-base-v2 (en_ec_sentence_transformer.py:22).  The MPNet encoder (relative-position
    attention bias shared by all layers, RoBERTa-style position ids, no token types, LayerNorm eps 1e-5) is restated in
    oracle/carel_oracle.py; this pins the restatement to the installed `transformers` MPNetModel (random init, 2 layers,
    dropout off) -- last hidden states and pooler output, with a padded batch.  (`sentence_transformers` itself stays
    unpinned: mean pooling + Normalize + the triplet loss are the published definitions.)"""
    transformers = pytest.importorskip("transformers")
    from transformers import MPNetConfig, MPNetModel
    torch.manual_seed(0)
    hc = MPNetConfig(vocab_size=300, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=514, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5,
                     relative_attention_num_buckets=32, pad_token_id=1)
    m = MPNetModel(hc, add_pooling_layer=True).eval()
    with torch.no_grad():
        m.encoder.relative_attention_bias.weight.mul_(25.0)            # default init (std 0.02) would hide a wrong bucket
    cfg = O.EncoderConfig(vocab_size=300, layers=2, max_pos=514, type_vocab=1, ln_eps=1e-5, variant="mpnet", pad_id=1, rel_pos=True)
    P = ST.params_from_mpnet_state_dict(m.state_dict(), cfg)
    g = torch.Generator().manual_seed(1)
    B, S = 5, 128
    ids = torch.randint(3, 300, (B, S), generator=g)
    lens = [128, 97, 33, 2, 64]
    att = torch.zeros((B, S), dtype=torch.long)
    for b, n in enumerate(lens):
        att[b, :n] = 1
        ids[b, n:] = 1                                                   # <pad> = 1 (position ids count non-pad tokens)
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=att)
        taps = {}
        pooled = O.encoder_forward(P, ids, att, torch.zeros_like(ids), cfg, taps=taps)
    for b, n in enumerate(lens):                                         # attended positions (HF leaves padded rows unspecified)
        a, r = taps["x2"][b, :n], ref.last_hidden_state[b, :n]
        assert float((a - r).abs().max()) < 2e-4 * max(1.0, float(r.abs().max())), b
    assert float((pooled - ref.pooler_output).abs().max()) < 2e-4
    # every distance bucket of S = 128 is the table row HF picks
    rp = O.mpnet_relative_position_bucket(torch.arange(-127, 128))
    assert int(rp.min()) == 0 and int(rp.max()) == 31 and rp[127] == 0 and rp[128] == 17 and rp[126] == 1
    want = m.encoder.compute_position_bias(torch.zeros(1, 16, 768))
    got = O.mpnet_position_bias(m.encoder.relative_attention_bias.weight.detach(), 16)
    assert torch.equal(want, got)
