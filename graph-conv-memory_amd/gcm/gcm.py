"""DenseGCM - the dense graph memory step (reference: src/gcm/gcm.py:151-355).

Same constructor, call surface and hidden-state layout as the reference:

    belief, m_t = DenseGCM(gnn, edge_selectors=...)(obs, m_t)
    m_t = (nodes f32[B,N,F], adj f32[B,N,N], weights f32[B,N,N] | f32[0], num_nodes i64[B])

What differs is where the work runs: the node insert + overflow roll, the edge
selectors, DenseGraphConv and the belief-row extract are gfx950 kernels
(include/gcm_hip.h) launched on the current stream, and the step never blocks
on the device: the reference's two host syncs per step (gcm.py:263, 316-318)
are replaced by a device-side flag word that is polled without blocking
(`finite_check="deferred"`, default), checked every step (`"sync"`, the
reference's exact raise point) or not at all (`"off"`).
"""
import math
import os
from typing import Tuple, Union

import weakref

import torch

from . import _hip, _ops


_DTYPES = (torch.float32, torch.float32, torch.float32, torch.float32, torch.int64, 1)


class PositionalEncoding(torch.nn.Module):
    """Embed a sin/cos positional encoding into the graph without touching future nodes
    (node index > num_nodes).  Reference: src/gcm/gcm.py:92-143 (same constructor, same lazily
    built `pe` buffer and `reproject` layer).  mode="add" is one in-place kernel
    (gcm_posenc_add); mode="cat" re-projects the features with gcm_rows_linear straight into the
    output's last columns and writes the first cat_dim columns from the table (gcm_posenc_cat_finish)."""

    def __init__(self, max_len: int = 5000, mode="add", cat_dim: int = 8):
        super().__init__()
        self.max_len = max_len
        self.mode = mode
        self.cat_dim = cat_dim
        assert mode in ["add", "cat"]

    def run_once(self, x: torch.Tensor) -> None:
        d_model = math.ceil(x.shape[-1] / 2) * 2          # gcm.py:103-112, table built on the host
        position = torch.arange(self.max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(self.max_len, d_model)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.to(x.device))
        if self.mode == "cat":
            self.reproject = torch.nn.Linear(x.shape[-1], x.shape[-1] - self.cat_dim,
                                             device=x.device)

    def forward(self, x: torch.Tensor, num_nodes: torch.Tensor) -> torch.Tensor:
        """x [B, N, F], num_nodes [B]: rows 0..num_nodes[b] (inclusive) are encoded."""
        if not hasattr(self, "pe"):
            self.run_once(x)
        if self.mode == "add":
            if x.is_leaf and x.requires_grad:
                x = x.clone()
            return _ops.posenc_add_(x.contiguous() if not x.is_contiguous() else x, self.pe,
                                    num_nodes)
        N, F = x.shape[1], x.shape[2]
        if x.is_cuda and x.dtype == torch.float32 and F <= 64 and 0 < self.cat_dim < F and x.dim() == 3 \
                and self.pe.shape[0] >= N:
            return _ops.posenc_cat(x, self.reproject.weight, self.reproject.bias, self.pe, num_nodes, self.cat_dim)
        # (wider features than the row kernel takes: torch ops)
        live = (torch.arange(N, device=x.device)[None, :] <= num_nodes[:, None]).unsqueeze(-1)
        pe = self.pe[:N, : self.cat_dim].unsqueeze(0).expand(x.shape[0], -1, -1)
        enc = torch.cat((pe, self.reproject(x)), dim=-1)
        return torch.where(live, enc, x)


class DenseGCM(torch.nn.Module):
    """Graph Associative Memory (dense adjacency)."""

    did_warn = False

    def __init__(
        self,
        gnn: torch.nn.Module,
        preprocessor: torch.nn.Module = None,
        edge_selectors: torch.nn.Module = None,
        aux_edge_selectors: torch.nn.Module = None,
        graph_size: int = 128,
        pooled: bool = False,
        positional_encoder: torch.nn.Module = None,
        edge_weights: bool = False,
        finite_check: str = "deferred",
        poll_interval: int = 16,
        mutate_num_nodes_on_overflow: bool = False,
        fused: bool = True,
        donate_state: bool = False,
    ):
        super().__init__()
        assert finite_check in ("deferred", "sync", "off")
        self.preprocessor = preprocessor
        self.gnn = gnn
        self.graph_size = graph_size
        self.edge_selectors = edge_selectors
        self.aux_edge_selectors = aux_edge_selectors
        self.pooled = pooled
        self.edge_weights = edge_weights
        self.positional_encoder = positional_encoder
        self.finite_check = finite_check
        self.poll_interval = poll_interval
        # gcm.py:354 decrements the CALLER's num_nodes in place when a graph wraps;
        # off by default (the returned values are identical either way)
        self.mutate_num_nodes_on_overflow = mutate_num_nodes_on_overflow
        # fused=True: when the GNN is the canonical two DenseGraphConv(+Tanh/ReLU) stack and the
        # selectors are native non-differentiable ones, a step runs as state-advance + selector
        # kernels + ONE fused GNN kernel under ONE autograd node (csrc/fused.hip)
        self.fused = fused
        # donate_state=True: the caller hands over the hidden state it passes in; the step advances
        # `nodes`, `adj` and `num_nodes` IN PLACE and returns the same tensors (no per-step clone of
        # the state, gcm.py:262,278,286).  Results are identical; a caller that keeps older hidden
        # states around (they would all alias the newest one) must leave this off.  Takes effect on
        # the live-row path (index-writing selectors, canonical GNN, no gradient w.r.t. obs / nodes);
        # every other case falls back to functional semantics.
        self.donate_state = donate_state
        self._plan_cache = None
        self._fold = None     # set by _structure(): what the live-row step absorbs besides the GNN
        self._noise_pool = None   # [exponential draws [noise_pool_steps, B, N], next index, drawn under capture] of the fused LearnedEdge step
        self._token = object()   # identifies hidden states produced by this module (_gcm_link)
        self._cfg_cache = {}
        self._cfg_last = None
        self._packed_cache = None
        self._flags = {}      # device -> uint32[1] flag word written by the kernels
        self._pending = []    # [(pinned host copy, event)] of flag words in flight
        self._pinned_pool = []
        self._ctr = [0]       # steps since the last poll (a list: nn.Module.__setattr__ is slow)
        self._fast = None     # (RowsFast, flag word) of the last live-row step: what __call__ tries first
        self._learned_chain = None   # (packed vector, LearnedChain, grad mode): the fused LearnedEdge steps' one autograd node
        # False: the round-2 backward of the fused LearnedEdge step (one kernel per step behind a [B,N,N]
        # gradient chain buffer) instead of the time-parallel one - kept for A/B tests
        self.learned_time_parallel = True
        # fused LearnedEdge steps: how many steps' gumbel draws one RNG launch makes (B x N x 4 bytes each)
        self.noise_pool_steps = 64
        # False: no cached steps (the first N steps of a LearnedEdge rollout from empty graphs on a donated state as
        # ONE launch each, the GNN behind the selection on per-chain caches) - A/B tests
        self.learned_cached_steps = True
        # ... a continuing LearnedEdge chain's steps run from DenseGCM.__call__ straight into the C++ host path
        # (LearnedFast in csrc/torch_ext/step_ext.cpp); False: every step through forward() - A/B tests
        self.learned_fast_path = True
        # False: no cached live-row steps (csrc/rows_cached.hip: the first N steps of a rollout from empty graphs on a
        # donated state, forward-only TemporalBackedge selectors: row cur alone over per-chain caches) - A/B
        self.rows_cached_steps = True
        # False: chains whose selectors also write COLUMN cur of the adjacency (DenseEdge; "backward" / "both" hops) stay
        # on the general live-row kernel instead of the column-write cached step (csrc/rows_colcache.hip: rank-1
        # updates of the chain's layer-1 aggregate, one matrix-core product of the live rows) - A/B tests
        self.rows_col_cache = os.environ.get("GCM_COL_CACHE", "1") == "1"
        # False: a cached EuclideanEdge chain as TWO launches per step (distance kernel, then the cached step) instead
        # of one (csrc/distance.hip: k_euclid_mfma2<.., TAIL>) - A/B tests; read when a chain is armed
        self.rows_one_launch_distance = True
        # True: the cached temporal-hops step (k_step_rows_cached_img4) reads the weight image in its interleaved form -
        # image2[layer][k][lane][rel | root], one 8-byte load per (W_rel[h][k], W_root[h][k]) pair straight into the
        # adjacent registers v_pk_fma_f32 wants: 64 load instructions instead of 128 and none of the 128 register moves
        # that paired the operands.  cfg2 on the same box, graph replay: 58.2 M against 55.2 M for the plain lane-major
        # image (False; env GCM_IMG_V4=0 for the A/B).  (Round 5's first use of this slot - four k per lane as 16-byte
        # loads - had measured slower: 4.85 against 4.63 us.)
        self.rows_weight_image_v4 = os.environ.get("GCM_IMG_V4", "1") == "1"
        # True (round 6): the cached temporal-hops step at F = H1 = 32 runs with a SECOND wave per graph that writes what
        # follows from `cur` alone (the observation into the node matrix / cache, the adjacency row, the count, the record's
        # live list) while wave 0 computes - ~120 instructions off a one-wave instruction stream.  False (env
        # GCM_ONE_WAVE=1): the one-wave kernel (A/B)
        self.rows_bookkeeping_wave = os.environ.get("GCM_ONE_WAVE", "0") != "1"
        # False: rollout() from empty graphs with forward temporal hops runs the persistent per-graph kernel of round 1
        # instead of the two-launch time-parallel forward (csrc/rollout_tp.hip) - A/B tests
        self.rollout_time_parallel = True
        # Steps whose observations / nodes need a gradient run on the live-row kernels too.  True: the whole
        # chain's dL/dx in ONE launch by the chain's single autograd node (hardware float atomics: the order of
        # summation, i.e. the last bits, is not fixed; a policy that feeds belief t-1 into observation t switches
        # to "steps" by itself); "steps": one light node and launch per step, ordered sums (bit-reproducible);
        # False: the round-1 fused kernels (one kernel per step and direction, full state saved) - A/B tests
        self.rows_dx = True

    # -- state ---------------------------------------------------------------
    def get_initial_hidden_state(self, x):
        """gcm.py:194-211 - zero (nodes, adj, weights, num_nodes) for a dummy x [B, feats]."""
        assert x.dim() == 2
        B, feats = x.shape
        edges = torch.zeros(B, self.graph_size, self.graph_size, device=x.device)
        nodes = torch.zeros(B, self.graph_size, feats, device=x.device)
        if self.edge_weights:
            weights = torch.zeros(B, self.graph_size, self.graph_size, device=x.device)
        else:
            weights = torch.zeros(0, device=x.device)
        num_nodes = torch.zeros(B, dtype=torch.long, device=x.device)
        return nodes, edges, weights, num_nodes

    def _fresh_state(self, x):
        """The zero state forward(x, None) starts from, as ONE allocation and ONE fill launch: nodes | adj | num_nodes are
        carved from it as independent tensors (own version counters, no view relation - what the C++ host path does for
        the functional state it returns), 256-byte aligned.  get_initial_hidden_state() itself keeps returning separately
        allocated tensors.  Three launches less per rollout from hidden = None (cfg2: 8 us of 457)."""
        if not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32 or self.edge_weights:
            return self.get_initial_hidden_state(x)
        B, feats = x.shape
        N = self.graph_size
        pad = lambda n: (n + 63) & ~63
        n_nodes, n_adj = pad(B * N * feats), pad(B * N * N)
        flat = torch.zeros(n_nodes + n_adj + 2 * B, device=x.device)
        st = flat.untyped_storage()
        nodes = torch.empty(0, device=x.device).set_(st, 0, (B, N, feats))
        edges = torch.empty(0, device=x.device).set_(st, n_nodes, (B, N, N))
        num_nodes = torch.empty(0, dtype=torch.long, device=x.device).set_(st, (n_nodes + n_adj) // 2, (B,))
        return nodes, edges, torch.zeros(0, device=x.device), num_nodes

    def rows_steps(self):
        """Number of steps this module has run on the live-row kernels (csrc/rows_step.hip) so far."""
        n = 0
        for cfg in self._cfg_cache.values():
            if cfg is not False and cfg._rows_fast is not None:
                n += cfg._rows_fast.steps()
        return n

    def learned_fast_steps(self):
        """Number of LearnedEdge steps that went from __call__ straight into the C++ host path (LearnedFast)."""
        n = 0
        for cfg in self._cfg_cache.values():
            if cfg is not False and cfg._learned_fast is not None:
                n += cfg._learned_fast.steps()
        return n

    def learned_steady_steps_taken(self):
        """Steps of the current LearnedEdge chain that ran as the one-launch steady-state step (gcm_learned_step_steady:
        past graph_size steps of a chain from empty graphs on a donated state)."""
        lc = self._learned_chain
        return int(lc[1].steady_steps()) if lc is not None else 0

    def rows_cached_steps_taken(self):
        """... of which cached steps (csrc/rows_cached.hip) in the chains armed last."""
        n = 0
        for cfg in self._cfg_cache.values():
            if cfg is not False and cfg._rows_fast is not None:
                n += cfg._rows_fast.cached_steps()
        return n

    def rows_col_steps_taken(self):
        """... of which column-write cached steps (gcm_dense_rows_step_colcache: DenseEdge / backward hops)."""
        n = 0
        for cfg in self._cfg_cache.values():
            if cfg is not False and cfg._rows_fast is not None:
                n += cfg._rows_fast.col_steps()
        return n

    def rows_rolled_steps_taken(self):
        """... of which steady-state ones (gcm_dense_rows_step_cached_roll: past N steps, every step drops the oldest node)."""
        n = 0
        for cfg in self._cfg_cache.values():
            if cfg is not False and cfg._rows_fast is not None:
                n += cfg._rows_fast.rolled_steps()
        return n

    def rows_cached_launches_per_step(self, B):
        """Kernel launches one cached step of this module's last configuration makes at batch size B (0: no cached
        form; 1: forward temporal hops, or EuclideanEdge alone as the one-launch form; 2: distance kernel + step) -
        gcm_dense_rows_cached_launches, a pure function of the configuration."""
        cfg = self._cfg_last[3] if self._cfg_last else None
        if cfg is None or not cfg.cpp_handle():
            return 0
        return cfg._cpp.cached_launches(B)

    def _flag_users(self):
        users = self.__dict__.get("_flag_user_list")
        if users is None:
            users = [m for sel in (self.edge_selectors, self.aux_edge_selectors) if sel is not None
                     for m in sel.modules() if hasattr(m, "_gcm_flags")]
            self.__dict__["_flag_user_list"] = users
        return users

    # -- device flag word ------------------------------------------------------
    def _flag_word(self, device):
        f = self._flags.get(device)
        if f is None:
            f = torch.zeros(1, dtype=torch.int32, device=device)
            self._flags[device] = f
        return f

    def _raise_for(self, bits):
        if bits & _hip.FLAG_BAD_COUNT:
            raise AssertionError("num_nodes outside [0, graph_size]")
        if bits & _hip.FLAG_WRAPPED and not DenseGCM.did_warn:
            print("Overflow detected, wrapping around. Will not warn again")
            DenseGCM.did_warn = True
        if bits & _hip.FLAG_WINDOW:
            from .edge_selectors.temporal import WINDOW_ERROR
            raise RuntimeError(WINDOW_ERROR.format("see TemporalBackedge.learning_window"))
        if bits & _hip.FLAG_NONFINITE:
            raise AssertionError("Got NaN in returned memory, try using tanh activation")

    def check_flags(self, block=True):
        """Surface anything the kernels flagged so far.  block=False only looks at
        flag copies that have already landed on the host."""
        if block:
            bits = 0
            for host, ev in self._pending:     # copies in flight carry bits the device word
                ev.synchronize()               # no longer has (it was zeroed behind each copy)
                bits |= int(host.item())
            self._pending.clear()
            for dev, f in self._flags.items():
                b = int(f.item())
                if b:
                    f.zero_()
                bits |= b
            self._raise_for(bits)
            return
        while self._pending and self._pending[0][1].query():
            host, ev = self._pending.pop(0)
            bits = int(host.item())
            self._pinned_pool.append((host, ev))
            self._raise_for(bits)

    def _poll(self, flags):
        c = self._ctr
        c[0] += 1
        if self.finite_check == "deferred" and c[0] < self.poll_interval:
            return                        # the common step: nothing to look at
        self._poll_due(flags)

    def _poll_due(self, flags):
        self._ctr[0] = 0
        if torch.cuda.is_current_stream_capturing():
            return                        # inside a HIP-graph capture: the flag word is read later
        if self.finite_check == "sync":
            bits = int(flags.item())
            if bits:
                flags.zero_()
            self._raise_for(bits)
        elif self.finite_check == "deferred":
            self.check_flags(block=False)
            self._enqueue_flag_copy(flags)

    def _enqueue_flag_copy(self, flags):
        """async copy of the flag word to a pinned host word (recycled), zeroing it behind"""
        pool = self._pinned_pool
        host, ev = pool.pop() if pool else (torch.empty(1, dtype=torch.int32, pin_memory=True),
                                            torch.cuda.Event())
        host.copy_(flags, non_blocking=True)
        flags.zero_()
        ev.record()
        self._pending.append((host, ev))
        if len(self._pending) > 64:       # nobody is polling: fold the oldest back
            h, e = self._pending.pop(0)
            e.synchronize()
            self._raise_for(int(h.item()))

    # -- fused fast path -----------------------------------------------------------
    def _structure(self):
        """(convs, acts, selector modules) when the module tree qualifies for the fused
        kernels, else None.  Structure is analysed once (modules are assumed static)."""
        if self._plan_cache is not None:
            return self._plan_cache[0]
        from . import nn as G
        from .edge_selectors.temporal import TemporalBackedge
        from .edge_selectors.dense import DenseEdge
        from .edge_selectors.distance import Distance

        def foldable():
            """What sits between the selectors and the GNN (gcm.py:290-306), when the live-row step
            can absorb it: a Linear preprocessor (folded into the layer-1 weights, its bias through
            the row sums of the adjacency), index-writing aux selectors, and - as in the reference only
            when aux selectors exist - the PositionalEncoding ("add": a table added to the rows <= cur
            of the GNN's input; "cat": seen by the aux selectors only, which do not read it)."""
            pre = self.preprocessor
            if isinstance(pre, torch.nn.Sequential) and len(pre) == 1:
                pre = pre[0]
            if pre is not None and type(pre) is not torch.nn.Linear:
                return None
            aux = self.aux_edge_selectors
            if aux is not None and not (isinstance(aux, DenseEdge) or
                                        (isinstance(aux, TemporalBackedge) and not aux.learned)):
                return None
            pe = self.positional_encoder if aux is not None else None   # gcm.py:294-301
            if pe is not None and type(pe) is not PositionalEncoding:
                return None
            if pe is not None and pe.mode == "add" and pre is not None:
                return None
            return {"pre": pre, "aux": aux, "pe": pe}

        def analyse():
            if not self.fused or self.pooled:
                return None
            self._fold = None
            if (self.preprocessor is not None or self.positional_encoder is not None
                    or self.aux_edge_selectors is not None):
                self._fold = foldable()
                if self._fold is None:
                    return None
            g = self.gnn
            if not isinstance(g, G.Sequential) or len(g.arg_names) < 2:
                return None
            xn, an = g.arg_names[0], g.arg_names[1]
            convs, acts = [], []
            for mod, ins, outs in g.stages():
                if isinstance(mod, G.DenseGraphConv):
                    if ins != [xn, an] or outs != [xn] or len(convs) == 2:
                        return None
                    convs.append(mod)
                    acts.append(_hip.ACT_NONE)
                elif type(mod) in G._FUSABLE and convs and ins == [xn] and outs == [xn] \
                        and acts[-1] == _hip.ACT_NONE:
                    acts[-1] = G._FUSABLE[type(mod)]
                else:
                    return None
            if len(convs) != 2 or convs[0].out_channels != convs[1].in_channels:
                return None
            native = (TemporalBackedge, DenseEdge, Distance)
            sel = self.edge_selectors
            from .edge_selectors.learned import LearnedEdge
            if isinstance(sel, LearnedEdge):
                # the fused learned step (csrc/learned_step.hip): default edge network only
                if (sel.deterministic or _ops.default_edge_network(sel.edge_network) is None
                        or self._fold is not None):
                    return None
                return convs, tuple(acts), [sel]
            if sel is None:
                mods = []
            elif isinstance(sel, native):
                mods = [sel]
            elif isinstance(sel, G.Sequential):
                mods = []
                names = sel.arg_names
                for mod, ins, outs in sel.stages():
                    if not isinstance(mod, native) or ins != names or outs != names[1:3]:
                        return None
                    mods.append(mod)
            else:
                return None
            if self._fold is not None:
                if any(isinstance(m, Distance) for m in mods):    # (they would need the dX path)
                    return None
                if self._fold["aux"] is not None:
                    mods = mods + [self._fold["aux"]]
            return convs, tuple(acts), mods

        self._plan_cache = (analyse(),)
        return self._plan_cache[0]

    def _fused_plan(self, nodes, adj, weights, F):
        """StepConfig for the fused kernels, or None when this call must take the layered path."""
        st = self._structure()
        if st is None or weights.numel() != 0 or adj.requires_grad or not nodes.is_cuda:
            return None
        last = self._cfg_last
        if last is not None and last[0] == nodes.shape[1] and last[1] == F and last[2] == nodes.device:
            return last[3]
        key = (nodes.shape[1], F, nodes.device)
        cached = self._cfg_cache.get(key)
        if cached is not None:
            return cached if cached is not False else None
        from .edge_selectors.distance import Distance
        from .edge_selectors.learned import LearnedEdge
        convs, acts, mods = st
        N, H1, H2 = nodes.shape[1], convs[0].out_channels, convs[1].out_channels
        cfg = False
        if mods and isinstance(mods[0], LearnedEdge):
            sel = mods[0]
            net = _ops.default_edge_network(sel.edge_network)
            # the kernels index the edge network with the fixed F-wide layout of learned.py:38-51
            # (2F -> F -> F -> 1, LayerNorms over F, every bias present): anything else - e.g. a hidden
            # width G != F, which default_edge_network accepts - takes the layered path
            if (convs[0].in_channels == F and net is not None and net[0].in_features == 2 * F
                    and net[0].out_features == F and net[3].in_features == F and net[3].out_features == F
                    and net[6].in_features == F and net[6].out_features == 1
                    and tuple(net[2].normalized_shape) == (F,) and tuple(net[5].normalized_shape) == (F,)
                    and all(net[i].bias is not None for i in (0, 3, 6))
                    and _hip.lib().gcm_learned_step_supported(N, F, H1, H2)):
                has_bias = (1 if convs[0].lin_rel.bias is not None else 0) | \
                           (2 if convs[1].lin_rel.bias is not None else 0)
                cfg = _ops.StepConfig([], acts, has_bias, N, F, H1, H2, nodes.device)
                cfg.convs = convs
                cfg.lins = (convs[0].lin_rel, convs[0].lin_root, convs[1].lin_rel, convs[1].lin_root)
                cfg.set_learned(sel, net)
        elif self._fold is not None:
            cfg = self._fold_config(convs, acts, mods, N, F, H1, H2, nodes.device)
        elif convs[0].in_channels == F and _ops.gnn2_supported(N, F, H1, H2):
            descs = [m.native_desc(F) if isinstance(m, Distance) else m.native_desc() for m in mods]
            if all(d is not None for d in descs):
                has_bias = (1 if convs[0].lin_rel.bias is not None else 0) | \
                           (2 if convs[1].lin_rel.bias is not None else 0)
                cfg = _ops.StepConfig(descs, acts, has_bias, N, F, H1, H2, nodes.device)
                cfg.convs = convs
                cfg.desc_sources = [m.pointer_source if isinstance(m, Distance) else None for m in mods]
                cfg.sharded = [m for m in mods if isinstance(m, Distance) and m.shard_group is not None]
                cfg.lins = (convs[0].lin_rel, convs[0].lin_root, convs[1].lin_rel, convs[1].lin_root)
        self._cfg_cache[key] = cfg
        if cfg is not False:
            self._cfg_last = (nodes.shape[1], F, nodes.device, cfg)
        return cfg if cfg is not False else None

    def _fold_config(self, convs, acts, mods, N, F, H1, H2, device):
        """StepConfig of a module with a folded preprocessor / positional encoding / aux selectors
        (live-row step only), or False."""
        fold = self._fold
        pre, pe = fold["pre"], fold["pe"]
        Fg = pre.out_features if pre is not None else F
        if (pre is not None and pre.in_features != F) or convs[0].in_channels != Fg:
            return False
        if pe is not None and not hasattr(pe, "pe"):      # the lazily built table / reproject layer
            pe.run_once(torch.empty(0, Fg, device=device))
        pe_add = pe is not None and pe.mode == "add"
        if pe_add and (pe.pe.shape[0] < N or pe.pe.shape[1] < F):
            return False
        descs = [m.native_desc() for m in mods]
        if any(d is None for d in descs):
            return False
        has_bias = (1 if convs[0].lin_rel.bias is not None else 0) | \
                   (2 if convs[1].lin_rel.bias is not None else 0)
        if pre is not None and pre.bias is not None:
            has_bias |= _hip.GNN_HAS_DEG_TERM
        if pe_add:
            has_bias |= _hip.GNN_HAS_PE_TABLE
        cfg = _ops.StepConfig(descs, acts, has_bias, N, F, H1, H2, device)
        if not cfg.rows_ok:
            return False
        cfg.convs = convs
        cfg.desc_sources = [None] * len(descs)
        cfg.lins = (convs[0].lin_rel, convs[0].lin_root, convs[1].lin_rel, convs[1].lin_root)
        cfg.fold = (pre, pe if pe_add else None)
        cfg.dx_ok = False      # (a gradient w.r.t. the observations through a folded transform: the layered path)
        return cfg

    def _packed_params(self, cfg, head=False):
        """The six GNN tensors as one flat vector (layout of include/gcm_hip.h "packed parameter
        vector"), so that a step returns ONE gradient tensor.  Rebuilt at the head of every chain
        of hidden states (`head`: hidden is None or not produced by this module - the reference
        reads live parameters, and writes through `.data` / `module.float()` bump no version
        counter), when a parameter changed, or after the previous vector took part in a backward
        pass; reused only inside a linked chain."""
        # current parameter tensors through the modules' own dicts (nn.Module.__getattr__ chains
        # cost ~0.5 us each and this runs every step)
        (rel0, root0, rel1, root1) = cfg.lins
        tensors = (rel0._parameters["weight"], root0._parameters["weight"], rel0._parameters["bias"],
                   rel1._parameters["weight"], root1._parameters["weight"], rel1._parameters["bias"])
        if cfg.learned_sel is not None:     # ... | edge network (layout of gcm_learned_* in gcm_hip.h)
            l0, _, n0, l1, _, n1, l2 = cfg.mlp_mods
            tensors = tensors + (l0.weight, l0.bias, n0.weight, n0.bias, l1.weight, l1.bias, n1.weight,
                                 n1.bias, l2.weight, l2.bias)
        fold = cfg.fold
        if fold is not None and fold[0] is not None:
            tensors = tensors + (fold[0]._parameters["weight"], fold[0]._parameters["bias"])
        cache = self._packed_cache
        if cache is not None and not head and not cache[2][0] and cache[0] == torch.is_grad_enabled():
            # valid while the very same tensor objects have not been written to (optimizer steps
            # bump _version; re-assigned Parameters are new objects)
            live = True
            for t, (t0, v0) in zip(tensors, cache[3]):
                if t is not t0 or (t is not None and t._version != v0):
                    live = False
                    break
            if live:
                return cache[1]
        # drop the previous vector FIRST: its graph keeps the parameters' AccumulateGrad nodes alive,
        # bound to the stream they were created on - a later HIP-graph capture of this module would
        # then be made to synchronise with that (possibly the default) stream and fail
        self._packed_cache = cache = None
        self._learned_chain = None
        if cfg._rows_fast is not None:
            cfg._rows_fast.forget()
        if cfg._learned_fast is not None:
            cfg._learned_fast.forget()
        key = torch.is_grad_enabled()
        sizes = (cfg.H1 * cfg.F, cfg.H1 * cfg.F, cfg.H1, cfg.H2 * cfg.H1, cfg.H2 * cfg.H1, cfg.H2)
        if cfg.learned_sel is not None:
            Fe = cfg.F
            sizes = sizes + (2 * Fe * Fe, Fe, Fe, Fe, Fe * Fe, Fe, Fe, Fe, Fe, 1)
        dev = tensors[0].device
        ext = _ops._ext.module()
        if fold is None and ext is not None and hasattr(ext, "pack_params") and all(t is not None for t in tensors):
            packed = ext.pack_params(list(tensors))     # (one autograd node: every parameter's gradient a view)
        else:
            parts = [t.reshape(-1) if t is not None else torch.zeros(n, device=dev)
                     for t, n in zip(tensors, sizes)]
            if fold is not None:
                parts = self._folded_parts(cfg, tensors, parts)
            packed = torch.cat(parts)
        if cfg.learned_sel is not None:
            assert packed.numel() == cfg.P_total, "edge network does not have the fused kernels' layout"
        used = [False]
        if packed.requires_grad:
            packed.register_hook(lambda g: used.__setitem__(0, True))
        # (the gate of the per-step fused nodes - _gate() - is made when one of them first asks for it)
        gate = [None, None, cfg.P_total if cfg.learned_sel is not None else cfg.P, dev]
        for d in cfg.descs:               # a re-assigned dist_param must not leave a stale pointer
            if d.kind == _hip.SEL_DISTANCE:
                cfg.refresh_pointers()
                break
        self._packed_cache = (key, packed, used,
                              [(t, t._version if t is not None else 0) for t in tensors], gate)
        return packed

    def _gate(self):
        """(gated packed vector, slab holder) of the current packed vector - what the per-step fused nodes consume
        the parameters through in grad mode (_ops._ParamGate: their parameter-gradient slabs summed once) - or
        (None, None) without gradients.  Made on first use: the live-row and rollout paths never ask."""
        pc = self._packed_cache
        g = pc[4]
        if g[0] is None and pc[1].requires_grad:
            g[1] = _ops.SlabHolder(g[2], g[3])
            g[0] = _ops.param_gate(pc[1], g[1])
        return g[0], g[1]

    @staticmethod
    def _folded_parts(cfg, tensors, parts):
        """Packed vector of a folded configuration (include/gcm_hip.h, "Folded node transforms"):
        layer 1 composed with the Linear preprocessor x' = W_p x + b_p -
            W_rel1 W_p | W_root1 W_p | b1 + W_root1 b_p | layer 2 | c1 = W_rel1 b_p | pe table
        built with torch ops, so autograd carries the kernel's gradient back to W_p, b_p and the
        original layer-1 tensors."""
        pre, pe = cfg.fold
        if pre is not None:
            w_rel, w_root, b1 = tensors[0], tensors[1], tensors[2]
            w_p, b_p = tensors[-2], tensors[-1]
            parts[0] = (w_rel @ w_p).reshape(-1)
            parts[1] = (w_root @ w_p).reshape(-1)
            if b_p is not None:
                parts[2] = parts[2] + w_root @ b_p
                parts.append(w_rel @ b_p)
        if pe is not None:
            parts.append(pe.pe[: cfg.N, : cfg.F].reshape(-1))
        return parts

    @staticmethod
    def _gather_sharded(cfg, x):
        """EuclideanEdge(shard_group=...): every rank's current nodes (= the observations going in) into
        the selector's persistent buffer, ahead of the kernels that read it (SURVEY 8e "Exception")."""
        for m in cfg.sharded:
            if m.gather_current(x):
                cfg.refresh_pointers()

    def _forward_rows(self, x, hidden, cfg, flags, link, need_dx=0):
        """The live-row step (csrc/rows_step.hip), checked entry: one kernel forward, no kernel and no
        autograd node per step backward (every step of a chain hangs its belief tensor on one node,
        whose backward is one time-parallel launch over every recorded step).  Taken when neither x
        nor the incoming node matrix needs a gradient.  A continuing chain does not come through
        here at all: DenseGCM.__call__ hands it to the C++ host path (RowsFast.step) directly."""
        nodes, adj, weights, num_nodes = hidden
        fast = cfg.rows_fast(self)
        if need_dx and fast.edited_dx_state(nodes, adj, weights, num_nodes):
            # The chain's dL/dx follows the rows of the node matrix back to the steps that inserted them BY POSITION;
            # a state the caller has edited in place (zeroed graphs of finished episodes ...) no longer has that form.
            if fast.donates():
                raise RuntimeError("donate_state=True with observation gradients: the hidden state was modified in "
                                   "place between two steps of a chain - use functional state (donate_state=False), "
                                   "or detach the chain (start it again from the edited state under a new call)")
            self._fast = None
            return self._forward_fused(x, nodes, adj, weights, num_nodes, cfg, flags, link)   # (full state saved)
        cont = fast.continues(nodes, adj, weights, num_nodes)
        root = self._packed_params(cfg, head=link is None and not cont)
        if not x.is_contiguous():
            x = x.contiguous()
        if not (nodes.is_contiguous() and adj.is_contiguous() and num_nodes.is_contiguous()):
            if self.donate_state:
                raise ValueError("donate_state=True needs contiguous hidden-state tensors")
            nodes, adj, num_nodes = nodes.contiguous(), adj.contiguous(), num_nodes.contiguous()
        # (a donated state with a gradient w.r.t. the observations: only in the one-node form, and the returned
        #  node matrix is then a plain tensor - RowsFast::want_donate decides and reports)
        fresh = getattr(nodes, "_gcm_fresh", False)
        if fresh:
            nodes._gcm_fresh = False          # (a donated state is this very tensor at every later step)
            if cfg.cpp_handle():
                cfg._cpp.set_cached_flags(
                    (0 if (self.rows_one_launch_distance or not cfg.has_distance) else _hip.STEP_TWO_LAUNCH)
                    | (_hip.STEP_IMG_V4 if self.rows_weight_image_v4 else 0)
                    | (0 if self.rows_bookkeeping_wave else _hip.STEP_ONE_WAVE))
                cfg._cpp.set_col_cache(bool(self.rows_col_cache))
        mx, n2, a2, c2, donate = fast.run(x, nodes, adj, weights, num_nodes, root, flags, cfg.cpp_handle(),
                                          self.donate_state, need_dx, bool(fresh and self.rows_cached_steps))
        if donate:
            out = hidden
        else:
            if self.mutate_num_nodes_on_overflow:
                num_nodes.copy_(c2 - 1)
            out = (n2, a2, weights, c2)
        # the next call of this module tries the unchecked entry first (not with the compat flag that
        # writes into the caller's num_nodes: that needs the checked path every step)
        if self.mutate_num_nodes_on_overflow or cfg.sharded:   # (a collective ahead of every step)
            self._fast = None
        else:    # (bound C++ entry, flag word, steps between looks at it)
            every = {"deferred": self.poll_interval, "sync": 1, "off": float("inf")}[self.finite_check]
            self._fast = (fast.step, flags, every)
        if self.finite_check != "off":
            self._poll(flags)
        return mx, out

    def _forward_learned(self, x, hidden, cfg, flags, link):
        """DenseGCM + LearnedEdge as fused kernels (csrc/learned_step.hip): 4 launches forward, one
        backward.  The returned adjacency carries no autograd history of its own: its gradient chain
        travels with the hidden state in compact form (`_gcm_dchain` on the adjacency tensor)."""
        nodes, adj, weights, num_nodes = hidden
        root = self._packed_params(cfg, head=link is None)
        B = x.shape[0]
        sel = cfg.learned_sel
        if sel.noise_fn is not None:      # injected gumbel draws (parity tests); values of the argument unspecified
            noise, is_exp = sel.noise_fn(torch.empty(B, cfg.N, device=x.device)), 0
        else:                             # torch.nn.functional.gumbel_softmax draws -log(Exp(1)) the same way
            # (noise_pool_steps steps' worth per draw: one RNG launch instead of one per step - and a refill is a
            #  step on the interpreter's path: 64 since round 6, the RNG launches were 36 of a cfg5 rollout's 550 us)
            # A pool drawn outside a HIP-graph capture is not used inside one (and vice versa): the
            # captured kernels would keep reading its address after it has been replaced.
            pool, cap = self._noise_pool, torch.cuda.is_current_stream_capturing()
            if (pool is None or pool[1] >= pool[0].shape[0] or pool[2] != cap or pool[0].shape[1] != B
                    or pool[0].device != x.device):
                pool = self._noise_pool = [torch.empty(max(1, int(self.noise_pool_steps)), B, cfg.N,
                                                       device=x.device).exponential_(), 0, cap]
            noise, is_exp = pool[0][pool[1]], 1
            pool[1] += 1
        ext = _ops._ext.module()
        if self.learned_time_parallel and ext is not None and hasattr(ext, "learned_step2") and _ops.TIMER is None:
            # every step of a chain of hidden states hangs its belief tensor on ONE autograd node
            # (LearnedChainNode in csrc/torch_ext/step_ext.cpp); the adjacency carries the index of the step
            # that wrote it, so that the time-parallel backward can follow the chain (or tree) of states
            lc = self._learned_chain
            # (a chain that starts from the empty graphs of hidden = None runs its first N steps as cached steps,
            #  LearnedChain in step_ext.cpp: its caches belong to ONE such rollout)
            fresh = getattr(nodes, "_gcm_fresh", False)
            if fresh:
                nodes._gcm_fresh = False      # (a donated state is this very tensor at every later step)
            fresh = fresh and self.learned_cached_steps
            if (lc is None or lc[0] is not root or lc[1].executed() or lc[2] != torch.is_grad_enabled()
                    or (fresh and lc[1].total_steps() > 0)):
                lc = self._learned_chain = (root, ext.LearnedChain(cfg.learned_cpp_handle(), root,
                                                                   bool(self.donate_state)),
                                            torch.is_grad_enabled())
            lin = getattr(adj, "_gcm_lin", None)
            parent = lin[1] if (lin is not None and lin[0] is lc[1]) else -1
            mx, n2, a2, cur, c2, idx = ext.learned_step2(lc[1], x, nodes, adj, num_nodes, noise, is_exp, flags, parent,
                                                         bool(fresh))
            if idx >= 0:
                a2._gcm_lin = (lc[1], idx)
            # the next call of this chain may skip the interpreter: DenseGCM.__call__ -> LearnedFast.step (C++)
            if (sel.noise_fn is None and not self.mutate_num_nodes_on_overflow and hasattr(ext, "LearnedFast")
                    and self.learned_fast_path):
                lf = cfg.learned_fast(self)
                lf.arm(lc[1], self._noise_pool, self._token, weakref.ref(cfg), flags, root, weights, idx, B, x.shape[1],
                       cfg.N)
                every = {"deferred": self.poll_interval, "sync": 1, "off": float("inf")}[self.finite_check]
                self._fast = (lf.step, flags, every)
            if lc[1].donates():      # the state was advanced in place: the caller's own tuple
                n2._gcm_link = (self._token, a2, cfg, flags, None, root, x.shape, weights, c2)
                if self.mutate_num_nodes_on_overflow:
                    raise NotImplementedError("mutate_num_nodes_on_overflow with donate_state")
                if self.finite_check != "off":
                    self._poll(flags)
                return mx, hidden
        elif self._gate()[0] is not None:
            gated, holder = self._gate()
            is_head = link is None or link[5] is not root
            dchain = getattr(adj, "_gcm_dchain", None) if not is_head else None
            if dchain is None:
                dchain = cfg.zero_chain(B, x.device)
            mx, dchain_out, n2, a2, cur, c2 = _ops.learned_step(
                gated, dchain, x, nodes, adj, num_nodes, noise, is_exp, flags, cfg, holder.get(B), is_head)
            a2._gcm_dchain = dchain_out
        else:
            with torch.no_grad():
                mx, _, n2, a2, cur, c2 = _ops.learned_step(
                    root, cfg.zero_chain(B, x.device), x, nodes, adj, num_nodes, noise, is_exp, flags, cfg,
                    None, True)
        n2._gcm_link = (self._token, a2, cfg, flags, None, root, x.shape, weights, c2)
        if self.mutate_num_nodes_on_overflow:
            num_nodes.copy_(cur)
        if self.finite_check != "off":
            self._poll(flags)
        return mx, (n2, a2, weights, c2)

    def _forward_fused(self, x, nodes, adj, weights, num_nodes, cfg, flags, link=None):
        root = self._packed_params(cfg, head=link is None)
        gated, holder = self._gate()
        # In grad mode the steps consume the parameter vector through a gate node and accumulate
        # their parameter-gradient slabs into ONE array of the module (summed once by the gate,
        # _ops._ParamGate) instead of returning T gradients for the engine to add.
        # `_gcm_link` on the returned node matrix lets the next call skip re-validating a hidden
        # state this module produced itself, and tells whether it continues the same chain.
        if gated is not None:
            is_head = link is None or link[5] is not root
            packed, slab_acc = gated, holder.get(x.shape[0])
        else:
            is_head, packed, slab_acc = True, root, None
        fast = cfg.cpp_call() if _ops.TIMER is None else None
        if fast is not None:      # (function, handle, device index) of the C++ node: no wrapper layers
            mx, nodes_out, adj_out, cur, num_nodes_next = fast[0](
                x, nodes, packed, adj, num_nodes, flags, fast[1],
                torch._C._cuda_getCurrentRawStream(fast[2]), slab_acc, is_head)
        else:
            mx, nodes_out, adj_out, cur, num_nodes_next = _ops.fused_step(
                x, nodes, packed, adj, num_nodes, flags, cfg, slab_acc, is_head)
        nodes_out._gcm_link = (self._token, adj_out, cfg, flags, None, root, x.shape, weights,
                               num_nodes_next)
        if self.mutate_num_nodes_on_overflow:
            num_nodes.copy_(cur)
        if self.finite_check != "off":
            self._poll(flags)
        return mx, (nodes_out, adj_out, weights, num_nodes_next)

    def rollout(self, obs, hidden=None, batch_first=False, truncate=True):
        """T memory steps at once: obs [T, B, feat] -> (beliefs [T, B, H], hidden after the
        last step); batch_first=True: obs [B, T, feat] -> beliefs [B, T, H] - the shape RLlib's wrapper holds
        (ray_gcm.py:186-209: `flat` [B, T, F] in, the stacked beliefs [B*T, H] out).  The VALUES (beliefs, hidden
        state) are those of T calls of forward(), and so is every gradient that flows inside the call.  Index-writing
        selectors on the canonical GNN: the whole rollout is enqueued by one C call and is one autograd node;
        EuclideanEdge alone and LearnedEdge from empty graphs: time-parallel forwards (csrc/euclid_tp.hip,
        gcm_learned_rollout_fwd); every other configuration runs the per-step kernels in a loop - on a state this
        call owns, so the steps advance it IN PLACE whatever `donate_state` says (the caller never sees the
        intermediate states; the incoming state is copied once).

        truncate (LearnedEdge only - the one selector whose ADJACENCY carries a gradient, learned.py:89-113): True
        (default) - the hidden state going in and the one coming out are plain tensors: truncated BPTT at the call
        boundary, which is what RLlib's state passing does (ray_gcm.py:186-209 hands states over as detached
        batches).  A later call's loss then does not reach this call's edge selections through the returned
        adjacency, as it would across T forward() calls.  False: the chain of hidden states is kept across the call
        boundary exactly as T forward() calls keep it (functional state, per-step kernels; slower)."""
        assert obs.dim() == 3 and obs.dtype == torch.float32
        if batch_first:
            out, hidden = self.rollout(obs.transpose(0, 1), hidden, truncate=truncate)
            return out.transpose(0, 1), hidden
        if obs.is_cuda and obs.device.index != torch.cuda.current_device():
            with torch.cuda.device(obs.device):
                return self.rollout(obs, hidden, truncate=truncate)
        if not truncate and torch.is_grad_enabled():
            from .edge_selectors.learned import LearnedEdge
            if any(isinstance(m, LearnedEdge) for sel in (self.edge_selectors, self.aux_edge_selectors)
                   if sel is not None for m in sel.modules()):
                outs = []
                for t in range(obs.shape[0]):        # T forward() calls: the chain (_gcm_lin / adj.grad_fn) survives
                    mx, hidden = self(obs[t], hidden)
                    outs.append(mx)
                if not outs:
                    return self.rollout(obs, hidden)
                return torch.stack(outs), hidden
        if obs.shape[0] == 0:
            if hidden is None:
                hidden = self.get_initial_hidden_state(obs[0] if obs.shape[0] else obs.new_zeros(obs.shape[1:]))
            return obs.new_zeros(0, obs.shape[1], 0), hidden
        fresh = hidden is None
        if fresh and not self.edge_weights:
            # (the plan looks at shapes and devices only: the 21 MB empty state is materialised by whoever needs it)
            z = obs.new_empty(0, self.graph_size, obs.shape[-1])
            cfg = self._fused_plan(z, obs.new_empty(0, self.graph_size, self.graph_size), obs.new_zeros(0), obs.shape[-1])
            # (T <= N: the adjacency entries a step writes are read by later steps' layer 1 - their gradient reaches
            #  the selection through ONE chain's records, so a rollout is not split at the overflow)
            if (cfg is not None and cfg.learned_sel is not None and self.learned_time_parallel
                    and self.learned_cached_steps and obs.shape[0] <= self.graph_size
                    and not (torch.is_grad_enabled() and obs.requires_grad)):
                return self._rollout_learned(obs, cfg)
            # forward temporal hops only, from empty graphs, observations without gradient: no recurrence at all - the
            # whole forward as two launches over every (step, graph) (csrc/rollout_tp.hip), one autograd node
            if (cfg is not None and cfg.learned_sel is None and cfg.fold is None and cfg.rows_ok
                    and self.rollout_time_parallel
                    and not (torch.is_grad_enabled() and obs.requires_grad)):
                ext = _ops._ext.module()
                if ext is not None and hasattr(ext, "rows_rollout_tp") and cfg.cpp_handle():
                    flags = self._flag_word(obs.device)
                    r = ext.rows_rollout_tp(cfg.cpp_handle(), self._packed_params(cfg, head=True), obs, flags)
                    if r is not None:
                        mx, nodes, adj, count = r
                        if self.finite_check == "sync":
                            self.check_flags()
                        elif self.finite_check == "deferred" and not torch.cuda.is_current_stream_capturing():
                            self.check_flags(block=False)
                            self._enqueue_flag_copy(flags)
                        if obs.shape[0] > self.graph_size and self.finite_check != "off":
                            flags.bitwise_or_(_hip.FLAG_WRAPPED)     # gcm.py:264-266: the one-time overflow notice
                        return mx, (nodes, adj, torch.zeros(0, device=obs.device), count)
        if fresh:
            hidden = self.get_initial_hidden_state(obs[0])
        nodes, adj, weights, num_nodes = hidden
        cfg = self._fused_plan(nodes, adj, weights, obs.shape[-1])
        if cfg is not None and (cfg.learned_sel is not None or cfg.fold is not None):
            cfg = None                    # LearnedEdge / folded transforms: the per-step kernels, in a loop
        elif (cfg is not None and cfg.has_distance and cfg.rows_ok
              and not (torch.is_grad_enabled() and (obs.requires_grad or nodes.requires_grad))):
            cfg = None                    # distance selectors: selector kernel + live-row step per step
        if cfg is None:
            return self._rollout_loop(obs, hidden, fresh)
        B, N = obs.shape[1], nodes.shape[1]
        assert (nodes.shape[0], adj.shape, num_nodes.shape[0], nodes.shape[2]) == \
            (B, (B, N, N), B, obs.shape[2]), "hidden state and observation shapes disagree"
        flags = self._flag_word(obs.device)
        mx_all, nodes_T, adj_T, count_T = _ops.fused_rollout(
            obs, nodes, self._packed_params(cfg, head=True), adj, num_nodes, flags, cfg)
        if self.finite_check == "sync":
            self.check_flags()
        elif self.finite_check == "deferred" and not torch.cuda.is_current_stream_capturing():
            self.check_flags(block=False)      # gcm.py:316-318 for rollout-only loops
            self._enqueue_flag_copy(flags)
        return mx_all, (nodes_T, adj_T, weights, count_T)

    def _rollout_learned(self, obs, cfg):
        """rollout() with LearnedEdge from hidden = None, T <= graph_size steps, observations without gradient: TWO
        launches (gcm_learned_rollout_fwd: the selection of a step depends on raw observations and its gumbel draws
        only, so every (graph, step) is a workgroup of one launch; the beliefs follow in a second one) and one
        autograd node whose backward is the chain's time-parallel one.  The hidden state it returns is a plain
        tensor tuple: like every rollout() it ends its gradient chain (a later call's gradient does not reach this
        rollout's selections - truncated BPTT at the call boundary, which is what RLlib's state passing does)."""
        ext = _ops._ext.module()
        T, B = obs.shape[0], obs.shape[1]
        N = cfg.N
        if (ext is None or not hasattr(ext, "learned_rollout") or not cfg.learned_cpp_handle() or (N & 3) or (cfg.F & 3)
                or B > 65535):            # (gcm_learned_rollout_fwd: GCM_EUNSUPPORTED above 65535 graphs - ADVICE r4)
            return self._rollout_loop(obs, self.get_initial_hidden_state(obs[0]), True)
        root = self._packed_params(cfg, head=True)
        sel = cfg.learned_sel
        if sel.noise_fn is not None:      # injected gumbel draws (parity tests): one call per step, in step order
            noise = torch.stack([sel.noise_fn(torch.empty(B, N, device=obs.device)) for _ in range(T)])
            is_exp = 0
        else:                             # torch.nn.functional.gumbel_softmax draws -log(Exp(1)) the same way
            noise, is_exp = torch.empty(T, B, N, device=obs.device).exponential_(), 1
        flags = self._flag_word(obs.device)
        mx, nodes, adj, count = ext.learned_rollout(cfg.learned_cpp_handle(), root, obs, noise, is_exp, flags)
        if self.finite_check == "sync":
            self.check_flags()
        elif self.finite_check == "deferred" and not torch.cuda.is_current_stream_capturing():
            self.check_flags(block=False)
            self._enqueue_flag_copy(flags)
        return mx, (nodes, adj, torch.zeros(0, device=obs.device), count)

    def _rollout_loop(self, obs, hidden, fresh):
        """rollout() as the loop of per-step calls.  The intermediate hidden states never leave this function, so -
        when no gradient flows through the state itself - the steps run on a state this call owns and advance it in
        place (the live-row / LearnedEdge kernels' donated form: no per-step copy of the 21 MB state), whatever the
        module's `donate_state`; a chain from hidden = None also gets the cached steps."""
        nodes = hidden[0]
        own = (not self.donate_state and self.fused and nodes.is_cuda and hidden[2].numel() == 0
               and not (torch.is_grad_enabled() and (obs.requires_grad or nodes.requires_grad or hidden[1].requires_grad)))
        if own:
            if fresh:
                hidden = None             # (the module's own empty graphs: the cached steps' precondition)
            else:
                hidden = (nodes.clone(), hidden[1].clone(), hidden[2], hidden[3].clone())
            self.donate_state = True
        try:
            outs = []
            for t in range(obs.shape[0]):
                mx, hidden = self(obs[t], hidden)
                outs.append(mx)
        finally:
            if own:
                self.donate_state = False
                self._fast = None         # (the C++ entry and the LearnedEdge chain were armed for a donated state)
                self._learned_chain = None
        return torch.stack(outs), hidden

    # -- the step --------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        """`belief, m = gcm(obs, m)`.  A step that continues the chain of the previous call on the
        live-row kernels is ONE call into the C++ host path (RowsFast.step validates exactly that:
        same hidden-state tensors as returned last, same observation shape, parameters untouched, grad
        mode unchanged, no module / global hooks registered); everything else goes through
        torch.nn.Module.__call__ and forward() below."""
        fast = self._fast
        if fast is not None and not kwargs and len(args) == 2:
            r = fast[0](args[0], args[1])
            if r is not None:
                c = self._ctr           # gcm.py:316-318, deferred: look at the flag word now and then
                c[0] += 1
                if c[0] >= fast[2]:
                    self._poll_due(fast[1])
                return r
        return torch.nn.Module.__call__(self, *args, **kwargs)

    def __getstate__(self):
        """copy.deepcopy / pickle of a module that has already run: the per-shape plans, the packed
        parameter vector and the C++ host path are runtime caches (rebuilt on the first call)."""
        d = dict(self.__dict__)
        d.update(_plan_cache=None, _fold=None, _noise_pool=None, _token=object(), _cfg_cache={}, _cfg_last=None,
                 _packed_cache=None, _flags={}, _pending=[], _pinned_pool=[], _ctr=[0], _fast=None,
                 _learned_chain=None, _flag_user_list=None)
        return d

    def forward(
        self,
        x,
        hidden: Union[None, Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]],
    ) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]:
        """x [B, feat]; hidden = (nodes [B,N,feat], adj [B,N,N], weights [B,N,N] | [0],
        num_nodes [B]) or None.  Returns (belief [B, H], new hidden)."""
        if hidden is None:
            hidden = self._fresh_state(x)
            hidden[0]._gcm_fresh = True     # empty graphs: what the cached LearnedEdge steps may rely on
        nodes, adj, weights, num_nodes = hidden

        # the kernels are launched on the CURRENT device's stream: tensors on another GPU get a
        # device guard for the call (PyTorch ops in the reference guard implicitly)
        if x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return self.forward(x, hidden)

        # A hidden state this module returned itself (same node / adjacency tensors, same input
        # shape): everything checked below held for it by construction.
        link = getattr(nodes, "_gcm_link", None)
        if (link is not None and link[0] is self._token and link[1] is adj and link[7] is weights
                and link[8] is num_nodes and x.shape == link[6] and x.dtype is torch.float32):
            cfg = link[2]
            no_dx = not (torch.is_grad_enabled() and (x.requires_grad or nodes.requires_grad))
            if cfg.sharded:
                self._gather_sharded(cfg, x)
            if cfg.rows_ok and (no_dx or (cfg.dx_ok and self.rows_dx)):
                return self._forward_rows(x, hidden, cfg, link[3], link, 0 if no_dx else (2 if self.rows_dx == "steps" else 1))
            if cfg.learned_sel is None and cfg.fold is None:
                return self._forward_fused(x, nodes, adj, weights, num_nodes, cfg, link[3], link)
            if no_dx:
                return self._forward_learned(x, hidden, cfg, link[3], link)
            # (observations with gradient: the layered path below)

        # gcm.py:246-260, as one comparison
        if (x.dtype, nodes.dtype, adj.dtype, weights.dtype, num_nodes.dtype, num_nodes.dim()) != _DTYPES:
            assert x.dtype == torch.float32
            assert nodes.dtype == torch.float
            assert adj.dtype == torch.float, "adj must be float32"
            assert weights.dtype == torch.float
            assert num_nodes.dtype == torch.long
            assert num_nodes.dim() == 1
        N = nodes.shape[1]
        B = x.shape[0]
        assert N == adj.shape[1] == adj.shape[2], "N must be equal for adj mat and node mat"

        flags = self._flag_word(x.device)
        plan = self._fused_plan(nodes, adj, weights, x.shape[-1])
        if plan is not None:
            # raw pointers from here on: the shapes must agree (the reference raises an indexing
            # error on a mismatched hidden state)
            assert (nodes.shape[0], adj.shape[0], num_nodes.shape[0], nodes.shape[2]) == \
                (B, B, B, x.shape[1]), "hidden state and observation shapes disagree"
            no_dx = not (torch.is_grad_enabled() and (x.requires_grad or nodes.requires_grad))
            if plan.sharded:
                self._gather_sharded(plan, x)
            if plan.rows_ok and (no_dx or (plan.dx_ok and self.rows_dx)):
                return self._forward_rows(x, hidden, plan, flags, None, 0 if no_dx else (2 if self.rows_dx == "steps" else 1))
            if plan.learned_sel is None and plan.fold is None:
                return self._forward_fused(x, nodes, adj, weights, num_nodes, plan, flags)
            if no_dx:
                return self._forward_learned(x, hidden, plan, flags, None)
        # insert x at row num_nodes (after the overflow roll); fresh nodes/adj/weights buffers
        nodes, adj, weights, cur, num_nodes_next = _ops.state_advance(
            nodes, adj, weights, num_nodes, x, flags)
        if self.mutate_num_nodes_on_overflow:
            num_nodes.copy_(cur)          # == num_nodes unless the graph wrapped (gcm.py:354)

        # The returned `nodes` must stay clean (gcm.py:275-278): hand user modules a copy
        # because they may write in place; native selectors/GNN layers never mutate nodes.
        user_code = self.preprocessor is not None or self.positional_encoder is not None
        dirty_nodes = nodes.clone() if user_code else nodes

        for m in self._flag_users():      # selectors whose kernels raise flags write this module's word
            m._gcm_flags = flags
        if self.edge_selectors:
            adj, weights = self.edge_selectors(dirty_nodes, adj, weights, cur, B)
        if self.preprocessor:
            dirty_nodes = self.preprocessor(dirty_nodes)
        if self.aux_edge_selectors:
            seen = dirty_nodes
            if self.positional_encoder:
                seen = self.positional_encoder(dirty_nodes, cur)
            adj, weights = self.aux_edge_selectors(seen, adj, weights, cur, B)

        node_feats = self.gnn(dirty_nodes, adj, weights, B, N)
        if self.pooled:
            mx = node_feats
            if self.finite_check != "off":   # gcm.py:316-318 on the pooled output
                flags.bitwise_or_((~torch.isfinite(mx)).any().to(torch.int32) * _hip.FLAG_NONFINITE)
        else:
            mx = _ops.gather_rows(node_feats, cur, flags)
        if self.finite_check != "off":
            self._poll(flags)
        return mx, (nodes, adj, weights, num_nodes_next)

    def wrap_overflow(self, nodes, adj, weights, num_nodes):
        """gcm.py:323-355 as a standalone call: returns rolled copies (inputs untouched,
        unless mutate_num_nodes_on_overflow)."""
        B, N, F = nodes.shape
        full = num_nodes + 1 > N
        x = torch.zeros(B, F, device=nodes.device)
        flags = self._flag_word(nodes.device)
        n2, a2, w2, cur, _ = _ops.state_advance(nodes, adj, weights, num_nodes, x, flags)
        # undo the insert of the dummy x for graphs that did not wrap
        keep = (~full).view(B, 1, 1)
        n2 = torch.where(keep, nodes, n2)
        if self.mutate_num_nodes_on_overflow:
            num_nodes.copy_(cur)
        return n2, a2, w2, cur
