"""CPU checks of the encoder's checker chain: the plain-torch restatement (tests/bert_ref.py) equals
`transformers.BertModel` on the synthetic weights, and the committed golden fixture is reproduced by it.
No device compute."""
import os

import numpy as np
import pytest
import torch

from tests import bert_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def blob():
    from shodh_memory_amd import build
    build.build()
    from shodh_memory_amd import embedder as E
    return E, E.synthetic_weights(1234)


def test_param_layout(blob):
    E, b = blob
    assert b.size == E.param_count() == 22_565_376           # BertModel(all-MiniLM-L6-v2) without the unused pooler
    sd = E.blob_to_state_dict(b)
    assert sd["encoder.layer.5.output.dense.weight"].shape == (384, 1536)
    assert abs(float(sd["embeddings.word_embeddings.weight"][:4].std()) - 0.2) < 0.03          # word table N(0, 0.2) ...
    assert abs(float(sd["encoder.layer.0.attention.self.query.weight"].std()) - 0.02) < 0.003   # ... everything else N(0, 0.02)
    assert (sd["embeddings.LayerNorm.weight"] == 1).all()
    assert np.array_equal(E.state_dict_to_blob(sd), b)
    assert np.array_equal(E.synthetic_weights(1234), b) and not np.array_equal(E.synthetic_weights(1235)[:100], b[:100])


def test_torch_restatement_matches_golden_and_transformers(blob):
    E, b = blob
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
    assert int(g["seed"]) == 1234
    sd = {k: torch.from_numpy(v.copy()) for k, v in E.blob_to_state_dict(b).items()}
    with torch.no_grad():
        for name in ("b1", "b4", "edge"):
            ids = torch.from_numpy(g[name + "_ids"].astype(np.int64)); mask = torch.from_numpy(g[name + "_mask"].astype(np.int64))
            emb = bert_ref.encode(sd, ids, mask).numpy()
            assert np.abs(emb - g[name + "_emb"]).max() < 2e-5, name     # golden came from transformers.BertModel
            n = np.linalg.norm(emb, axis=1)
            lens = g[name + "_mask"].sum(1)
            assert np.allclose(n[lens > 0], 1, atol=1e-5) and (n[lens == 0] == 0).all()
