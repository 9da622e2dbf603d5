#!/usr/bin/env python
"""Golden vectors for SURVEY 8f4: run the REAL reference batching helpers (/root/reference/deepblast/dataset/utils.py:
collate_f :255-281, pack_sequences :214-221, unpack_sequences :224-252) on synthetic items and store the inputs and
what they returned.  Data only; run in the build container (numba is replaced by oracle/_shim)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(1, "/root/reference")
import importlib.util  # noqa: E402

# deepblast/dataset/__init__.py pulls in Biopython (absent here); the helpers live in utils.py, which only needs
# torch, numpy, scipy and deepblast.constants: load that file on its own
_spec = importlib.util.spec_from_file_location("_ref_dataset_utils", "/root/reference/deepblast/dataset/utils.py")
_utils = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_utils)
collate_f, pack_sequences, unpack_sequences = _utils.collate_f, _utils.pack_sequences, _utils.unpack_sequences


def main():
    rng = np.random.default_rng(2024)
    sizes = [(5, 9), (12, 3), (1, 1), (7, 7), (12, 9), (3, 11)]
    batch, out = [], {"sizes": np.array(sizes, dtype=np.int64)}
    for b, (n, m) in enumerate(sizes):
        gene = torch.from_numpy(rng.integers(1, 21, n))
        other = torch.from_numpy(rng.integers(1, 21, m))
        states = torch.from_numpy(rng.integers(0, 3, n + m - 1))
        aln = torch.from_numpy(rng.random((n, m)).astype(np.float32))
        path = torch.from_numpy((rng.random((n, m)) * 4).astype(np.float32))
        mask = torch.from_numpy(rng.integers(0, 2, (n, m)))
        gm, om = torch.from_numpy(rng.integers(0, 2, n).astype(np.float32)), torch.from_numpy(rng.integers(0, 2, m).astype(np.float32))
        batch.append((gene, other, states, aln, path, mask, gm, om))
        for name, t in zip(("gene", "other", "states", "aln", "path", "mask", "gm", "om"), batch[-1]):
            out[f"i{b}_{name}"] = t.numpy()
    genes, others, states, dm, p, G, gM, oM = collate_f(batch)
    out.update(dm=dm.numpy(), p=p.numpy(), G=G.numpy(), gM=gM.numpy(), oM=oM.numpy())
    packed, order = pack_sequences(genes, others)
    x, xlen, y, ylen = unpack_sequences(packed, order)
    out.update(order=np.asarray(order), x=x.numpy(), xlen=xlen.numpy(), y=y.numpy(), ylen=ylen.numpy())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_batching.npz"), **out)
    print("dm", tuple(dm.shape), "xlen", xlen.tolist(), "ylen", ylen.tolist())


if __name__ == "__main__":
    main()
