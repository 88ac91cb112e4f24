#!/usr/bin/env python3
"""Generates tests/golden/encoder_int8_golden.npz from the numpy restatement of the dynamically quantised graph
(oracle/int8_ref.py) with the seed-1234 synthetic weights of libshodh_hip (host-only generator, no GPU needed):

    python tests/golden/make_int8_golden.py

Inputs: 4 token rows padded to 256 (lengths 5, 40, 128 and an EMPTY row). Outputs: pooled unit vectors, plus the quantisation
parameters and a checksum of the int32 accumulators of layer 0's query projection. NOTE: this pins the HIP INT8 mode to the
restated ONNX operator semantics, NOT to model_quint8_avx2.onnx (no ONNX Runtime / checkpoint offline): parity with the real file
is unpinned."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import int8_ref as R                     # noqa: E402
from shodh_memory_amd import embedder as E          # noqa: E402  (host-only helpers: synthetic blob, parameter names)


def inputs():
    rng = np.random.default_rng(77)
    lens = [5, 40, 128, 0]
    ids = np.zeros((4, 256), np.int32)
    mask = np.zeros((4, 256), np.uint8)
    for i, n in enumerate(lens):
        if n:
            ids[i, :n] = rng.integers(1000, 30522, n)
            ids[i, 0] = 101; ids[i, n - 1] = 102
            mask[i, :n] = 1
    return ids, mask


def main():
    sd = E.blob_to_state_dict(E.synthetic_weights(1234))
    ids, mask = inputs()
    tr = {}
    emb = R.encode(sd, ids, mask, trace=tr)
    acc = tr["acc_q"].astype(np.int64)
    out = os.path.join(ROOT, "tests", "golden", "encoder_int8_golden.npz")
    np.savez_compressed(out, ids=ids, mask=mask, emb=emb.astype(np.float32), a_scale=np.float32(tr["a_scale"]), a_zp=np.int32(tr["a_zp"]),
                        w_scale=np.float32(tr["w_scale"]), acc_q_sum=np.int64(acc.sum()), acc_q_abs_sum=np.int64(np.abs(acc).sum()),
                        acc_q_corner=tr["acc_q"][:8, :8].astype(np.int32))
    print("wrote", out, emb.shape, float(np.linalg.norm(emb[0])))


if __name__ == "__main__":
    main()
