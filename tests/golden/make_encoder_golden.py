"""Generates tests/golden/encoder_golden.npz: token inputs -> pooled unit vectors computed by
`transformers.BertModel` (all-MiniLM-L6-v2 architecture: 6 layers, hidden 384, 12 heads, FFN 1536,
vocab 30522) holding the SYNTHETIC weights of seed 1234 (shodh_embedder_synthetic_weights; the blob is
regenerated from the seed wherever it is needed, it is not stored). The reference itself cannot be
run here (Rust + ONNX Runtime + downloaded weights): this pins the ARCHITECTURE, not the checkpoint.

    python tests/golden/make_encoder_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shodh_memory_amd import embedder as E          # noqa: E402  (host-only helpers, no GPU needed)
from tests import bert_ref                            # noqa: E402

SEED = 1234
HARSH_SEED = 4242


def hf_model(blob):
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                     max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in E.blob_to_state_dict(blob).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def run_cases(m, cases, seeds):
    out = {}
    with torch.no_grad():
        for name, lens in cases.items():
            ids, mask = bert_ref.synth_batch(len(lens), 256, seed=seeds[name], lengths=lens)
            hidden = m(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids)).last_hidden_state
            out[name + "_ids"] = ids.numpy().astype(np.int32)
            out[name + "_mask"] = mask.numpy().astype(np.uint8)
            out[name + "_emb"] = bert_ref.pool(hidden, mask).numpy().astype(np.float32)
    return out


def main():
    blob = E.synthetic_weights(SEED)
    out = run_cases(hf_model(blob), {"b1": [3], "b4": [3, 17, 128, 64], "edge": [1, 2, 128, 0, 50]}, {"b1": 11, "b4": 12, "edge": 13})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"), seed=np.int64(SEED), **out)
    print({k: v.shape for k, v in out.items()})
    # the HARSH weight set (tests/synth_weights.py: outlier channels, LayerNorm gains over two decades, heavy-tailed word table): the closest
    # thing to a trained checkpoint the offline image allows. Same architecture oracle (transformers.BertModel, fp32, CPU).
    from tests import synth_weights as SW
    hb = SW.harsh_blob(E, HARSH_SEED)
    hout = run_cases(hf_model(hb), {"h8": [5, 9, 23, 40, 64, 97, 128, 128]}, {"h8": 21})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_harsh_golden.npz"), seed=np.int64(HARSH_SEED), **hout)
    print({k: v.shape for k, v in hout.items()}, "harsh fixture; hidden-state scale check:", float(np.abs(hb).max()))


if __name__ == "__main__":
    main()
