"""Helpers shared by the parity tests: load a golden fixture, rebuild the
oracle objects it describes."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(z["meta"]))
        self.t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}

    def __getitem__(self, k):
        return self.t[k]

    def __contains__(self, k):
        return k in self.t

    def group(self, prefix):
        return {k[len(prefix):]: v for k, v in self.t.items() if k.startswith(prefix)}

    def h0(self):
        if "h0_nodes" not in self.t:
            return None
        w = self.t["h0_weights"] if "h0_weights" in self.t else torch.zeros(0)
        return (self["h0_nodes"], self["h0_adj"], w, self["h0_num_nodes"])


def oracle_selector(meta, sel_params=None, noise=None):
    """Oracle selector object for a fixture's meta block."""
    from oracle import dense as od

    kind = meta["selector"]
    if kind == "temporal":
        return od.TemporalBackedge(meta["hops"], meta["direction"])
    if kind == "dense":
        return od.DenseEdge()
    dist_param = None
    if meta.get("learned"):
        dist_param = sel_params["dist_param"]
    if kind == "euclid":
        return od.EuclideanEdge(meta["max_distance"], dist_param=dist_param)
    if kind == "cosine":
        return od.CosineEdge(meta["max_distance"], dist_param=dist_param)
    if kind == "spatial":
        return od.SpatialEdge(meta["max_distance"], slice(*meta["a"]), slice(*meta["b"]),
                              dist_param=dist_param)
    raise KeyError(kind)
