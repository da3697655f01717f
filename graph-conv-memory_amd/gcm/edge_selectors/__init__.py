"""Dense edge selectors of the DenseGCM step (plugin API #1 of SURVEY 8b):
`forward(nodes, adj, weights, num_nodes, B) -> (adj, weights)`.

temporal.TemporalBackedge, dense.DenseEdge            index writes, folded into the step kernels
distance.EuclideanEdge / CosineEdge / SpatialEdge     fused pairwise distance + threshold kernels
learned.LearnedEdge                                   candidate pairs + gumbel-softmax/STE kernels
                                                      around a user-replaceable edge network
"""
