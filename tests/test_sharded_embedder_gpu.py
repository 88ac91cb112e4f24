"""ShardedEmbedder (shodh_memory_amd/distributed.py): texts dealt to several encoder handles -- one per GPU in production, two or three handles on the one
GPU here -- and gathered in input order. SURVEY 8(e): "Encoder: pure data-parallel over texts, no collective". Every vector must be byte for byte what
ONE handle returns for the same call (per-text scope: a text's vector does not depend on its batch mates)."""
import numpy as np
import pytest

from .conftest import has_gpu

pytestmark = pytest.mark.gpu


def _batch(n, seed, max_len=256):
    rng = np.random.default_rng(seed)
    ids = np.zeros((n, max_len), np.int32)
    mask = np.zeros((n, max_len), np.uint8)
    for i in range(n):
        ln = int(rng.integers(2, 129)) if i % 7 else 0            # every seventh text is empty (a zero vector, minilm.rs:1123-1125)
        if ln:
            ids[i, :ln] = rng.integers(1000, 30521, ln)
            ids[i, 0] = 101; ids[i, ln - 1] = 102
            mask[i, :ln] = 1
    return ids, mask


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
@pytest.mark.parametrize("dtype_name,handles", [("int8", 2), ("bf16", 3), ("fp32", 2)])
def test_sharded_embedder_equals_one_handle(dtype_name, handles):
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd.distributed import ShardedEmbedder
    from shodh_memory_amd.embedder import MiniLMEmbedder
    dtype = {"int8": L.DTYPE_INT8, "bf16": L.DTYPE_BF16, "fp32": L.DTYPE_FP32}[dtype_name]
    one = MiniLMEmbedder(synthetic_seed=11, dtype=dtype, device=0)
    sh = ShardedEmbedder([MiniLMEmbedder(synthetic_seed=11, dtype=dtype, device=0) for _ in range(handles)])
    assert sh.dimension() == 384
    for n in (1, handles - 1 or 1, handles, 37, 200):               # fewer texts than handles, uneven shares, a real batch
        ids, mask = _batch(n, 100 + n)
        want = one.encode_ids(ids, mask, scope=L.QUANT_SCOPE_PER_TEXT)
        got = sh.encode_ids(ids, mask)                              # per-text scope by default
        assert got.shape == want.shape and got.tobytes() == want.tobytes(), (dtype_name, n)
        empty = np.where(mask.sum(axis=1) == 0)[0]
        assert not got[empty].any()                                 # empty texts keep their positions
    if dtype == L.DTYPE_INT8:
        # the batch scope is ONE function of the whole batch (ranges span it): not split, identical to the single handle's batch call
        ids, mask = _batch(24, 7)
        assert sh.encode_ids(ids, mask, scope=L.QUANT_SCOPE_BATCH).tobytes() == one.encode_ids(ids, mask, scope=L.QUANT_SCOPE_BATCH).tobytes()
    # an error in one share reaches the caller (a mask that is not a prefix of ones)
    ids, mask = _batch(10, 3)
    mask[9, 5] = 0
    with pytest.raises(L.ShodhError):
        sh.encode_ids(ids, mask)
    sh.close(); one.close()
