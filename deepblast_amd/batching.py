"""Padded-batch collation with a lengths side-channel.

The reference pads every pair of a batch to the longest one (`collate_f`, deepblast/dataset/utils.py:255-281)
and aligns over the full padded matrix, because its operator has no notion of per-pair sizes.  The decoders of
this package accept `lengths` ((B,2) int: n_b, m_b); these helpers produce that tensor at the two places where
the reference still knows the true sizes:

* `collate_with_lengths(batch)` -- `collate_f` with one more element in the returned tuple;
* `lengths_from_unpacked(xlen, ylen)` -- from the lengths `unpack_sequences` returns
  (deepblast/dataset/utils.py:224-252, used in NeuralAligner.forward, alignment.py:107-110).

Host-side data plumbing only; nothing here touches the GPU path.
"""
import torch


def lengths_from_unpacked(xlen, ylen):
    """(B,), (B,) sequence lengths -> (B,2) int32 `lengths` for decoder(theta, A, lengths)."""
    xlen = torch.as_tensor(xlen).reshape(-1)
    ylen = torch.as_tensor(ylen).reshape(-1)
    if xlen.shape != ylen.shape:
        raise ValueError(f"xlen and ylen must have the same length, got {tuple(xlen.shape)} and {tuple(ylen.shape)}")
    return torch.stack([xlen, ylen], dim=1).to(torch.int32)


def collate_with_lengths(batch):
    """Same padding as the reference's `collate_f` (utils.py:255-281) plus the true sizes.

    batch: list of (gene, other, states, alignment (n,m), path (n,m), mask (n,m), g_mask (n,), o_mask (m,)).
    -> (genes, others, states, dm (B,N,M), p (B,N,M), G (B,N,M) bool, gM (B,N), oM (B,M), lengths (B,2) int32)
    """
    genes = [x[0] for x in batch]
    others = [x[1] for x in batch]
    states = [x[2] for x in batch]
    B = len(batch)
    lengths = torch.tensor([[len(g), len(o)] for g, o in zip(genes, others)], dtype=torch.int32).reshape(B, 2)
    N, M = int(lengths[:, 0].max()), int(lengths[:, 1].max())
    dm = torch.zeros((B, N, M))
    p = torch.zeros((B, N, M))
    G = torch.zeros((B, N, M), dtype=torch.bool)
    gM = torch.zeros((B, N))
    oM = torch.zeros((B, M))
    for b, item in enumerate(batch):
        n, m = int(lengths[b, 0]), int(lengths[b, 1])
        dm[b, :n, :m] = item[3]
        p[b, :n, :m] = item[4]
        G[b, :n, :m] = torch.as_tensor(item[5]).bool()
        gM[b, :n] = item[6]
        oM[b, :m] = item[7]
    return genes, others, states, dm, p, G, gM, oM, lengths
