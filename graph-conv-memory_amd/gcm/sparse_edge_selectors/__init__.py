"""Sparse edge selectors of the SparseGCM step (plugin API #3 of SURVEY 8b):
`forward(nodes, T, taus, B) -> torch.sparse_coo adjacency (batch, sink, source)`.

temporal.TemporalEdge   closed-form count + fill kernels
learned.LearnedEdge     closed-form causal candidates, pair gather, segmented gumbel softmax
"""
