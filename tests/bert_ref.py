"""Plain PyTorch fp32 restatement of the MiniLM-L6 (BERT) forward + the reference's pooling
(minilm.rs:959-981, :846-878). Checker only. Pinned against `transformers.BertModel` in
tests/test_encoder_cpu.py (architecture oracle)."""
import math

import torch


def bert_forward(sd, ids, mask, heads=12, eps=1e-12):
    """sd: name -> tensor (HF BertModel names). ids [B,S] long, mask [B,S] {0,1}. Returns last hidden [B,S,H] fp32."""
    B, S = ids.shape
    H = sd["embeddings.word_embeddings.weight"].shape[1]
    dh = H // heads
    pos = torch.arange(S, device=ids.device)
    x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.position_embeddings.weight"][pos][None] \
        + sd["embeddings.token_type_embeddings.weight"][0][None, None]
    x = torch.nn.functional.layer_norm(x, (H,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    bias = (1.0 - mask.to(torch.float32))[:, None, None, :] * torch.finfo(torch.float32).min
    n_layers = len([k for k in sd if k.endswith("attention.self.query.weight")])
    for l in range(n_layers):
        p = "encoder.layer.%d." % l
        lin = lambda name, t: t @ sd[p + name + ".weight"].T + sd[p + name + ".bias"]   # noqa: E731
        q = lin("attention.self.query", x).view(B, S, heads, dh).transpose(1, 2)
        k = lin("attention.self.key", x).view(B, S, heads, dh).transpose(1, 2)
        v = lin("attention.self.value", x).view(B, S, heads, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + bias, dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(B, S, H)
        x = torch.nn.functional.layer_norm(lin("attention.output.dense", ctx) + x, (H,), sd[p + "attention.output.LayerNorm.weight"],
                                           sd[p + "attention.output.LayerNorm.bias"], eps)
        h = torch.nn.functional.gelu(lin("intermediate.dense", x))        # erf form
        x = torch.nn.functional.layer_norm(lin("output.dense", h) + x, (H,), sd[p + "output.LayerNorm.weight"],
                                           sd[p + "output.LayerNorm.bias"], eps)
    return x


def pool(hidden, mask):
    """masked mean-pool + NaN/Inf scrub + L2 normalise (minilm.rs:959-981, :846-878)."""
    m = (mask == 1).to(hidden.dtype)[:, :, None]
    s = (hidden * m).sum(1)
    cnt = m.sum(1)
    pooled = torch.where(cnt > 0, s / cnt.clamp(min=1), s)
    pooled = torch.nan_to_num(pooled, nan=0.0, posinf=0.0, neginf=0.0)
    norm = pooled.norm(dim=1, keepdim=True)
    return torch.where(norm > torch.finfo(torch.float32).eps, pooled / norm.clamp(min=1e-30), pooled)


def encode(sd, ids, mask):
    return pool(bert_forward(sd, ids, mask), mask)


def synth_batch(b, max_len=256, seed=0, lengths=None):
    """SURVEY 8d token inputs: lengths ~U[8,128], ids ~U[1000,30521], [CLS]=101 ... [SEP]=102, right-padded."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros((b, max_len), dtype=torch.int64)
    mask = torch.zeros((b, max_len), dtype=torch.int64)
    for i in range(b):
        n = int(lengths[i]) if lengths is not None else int(torch.randint(8, 129, (1,), generator=g))
        if n == 0:
            continue
        row = torch.randint(1000, 30522, (n,), generator=g)
        row[0] = 101
        if n > 1:
            row[-1] = 102
        ids[i, :n] = row
        mask[i, :n] = 1
    return ids, mask
