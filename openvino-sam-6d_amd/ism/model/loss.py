"""Drop-in for the hot-path classes of ISM/model/loss.py (same names / constructor arguments / forward signatures)."""
import torch
from torch import nn

from sam6d_hip import ism as _ism


class PairwiseSimilarity(nn.Module):
    """loss.py:21-44."""

    def __init__(self, metric="cosine", chunk_size=64):
        super().__init__()
        self.metric, self.chunk_size = metric, chunk_size

    def forward(self, query, reference):
        return _ism.pairwise_similarity(query.contiguous(), reference.contiguous())


class MaskedPatch_MatrixSimilarity(nn.Module):
    """loss.py:46-76 (compute_straight / compute_visible_ratio, the two methods the detector calls)."""

    def __init__(self, metric="cosine", chunk_size=64):
        super().__init__()
        self.metric, self.chunk_size = metric, chunk_size

    def compute_straight(self, query, reference):
        q, r = query.contiguous(), reference.contiguous()
        return _ism.patch_scores(_ism.patch_similarity(q, r), q)[0]

    def compute_visible_ratio(self, query, reference, thred=0.5):
        q, r = query.contiguous(), reference.contiguous()
        return _ism.patch_scores(_ism.patch_similarity(q, r), q, thred)[1]
