"""DynamicsPredictor with the reference's constructor / forward / state_dict surface
(src/dynamics/gnn/model.py:64-342), executing on the HIP engine.

The nn.Module tree exists only so that state_dict keys, load_state_dict, .to() and .eval() behave like the reference;
no torch op of it is ever run - forward() hands raw device pointers to the C-ABI (inference only, no autograd).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .context import Engine, ptr, current_stream, _require_gpu
from .graph import EdgeList


def _mlp3(n_in, n_hidden, n_out):
    # same parameter names as the reference's Encoder.model (indices 0, 2, 4 are the Linear layers)
    return nn.Sequential(nn.Linear(n_in, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden), nn.ReLU(),
                         nn.Linear(n_hidden, n_out), nn.ReLU())


class _Holder(nn.Module):
    pass


class DynamicsPredictor(nn.Module):
    def __init__(self, model_config, material_config, dataset_config, device):
        super().__init__()
        self.model_config = model_config
        self.material_config = material_config
        self.dataset_config = dataset_config
        self.device = torch.device(device)
        self.n_his = dataset_config["n_his"]
        self.nf_particle = model_config["nf_particle"]
        self.nf_relation = model_config["nf_relation"]
        self.nf_effect = model_config["nf_effect"]
        self.eps = 1e-6
        self.motion_clamp = 100
        self.num_materials = len(material_config["material_index"])
        assert self.num_materials == 1, "Only support single material."          # model.py:89
        material_params = material_config[dataset_config["materials"][0]]["physics_params"]
        material_dim = sum(1 for p in material_params if p["use"])               # model.py:92-95
        input_dim = (self.n_his * model_config["state_dim"] + self.n_his * model_config["offset_dim"] +
                     model_config["attr_dim"] + model_config["action_dim"] + model_config["density_dim"] +
                     material_dim)                                               # model.py:97-102
        rel_particle_dim = model_config["rel_particle_dim"]
        if rel_particle_dim == -1:
            rel_particle_dim = input_dim
        rel_input_dim = (rel_particle_dim * 2 + model_config["rel_attr_dim"] * 2 + model_config["rel_group_dim"] +
                         model_config["rel_distance_dim"] * self.n_his + model_config["rel_density_dim"])
        if model_config["offset_dim"] > 0:
            raise NotImplementedError                                            # model.py:181-182
        unsupported = (model_config["state_dim"] != 0 or model_config["density_dim"] != 0 or
                       rel_particle_dim != 0 or model_config["rel_density_dim"] != 0 or
                       model_config["attr_dim"] != 2 or model_config["action_dim"] != 3 or material_dim != 1 or
                       model_config["rel_attr_dim"] != 2 or model_config["rel_group_dim"] != 1 or
                       model_config["rel_distance_dim"] != 3)
        if unsupported or input_dim != 6 or self.n_his not in (4, 5) or rel_input_dim != 5 + 3 * self.n_his or \
                not (self.nf_particle == self.nf_relation == self.nf_effect == 150):
            raise NotImplementedError(
                "the HIP kernels are built for input_dim 6, nf 150 and n_his 4 (rel_input_dim 17: config/dynamics/"
                "{rope,granular,cloth,...}.yaml) or n_his 5 (rel_input_dim 20: softbody.yaml) - got "
                f"{input_dim}, {self.nf_effect}, {self.n_his}, {rel_input_dim}")
        self.input_dim, self.rel_input_dim = input_dim, rel_input_dim

        nf = self.nf_effect
        self.particle_encoder = _Holder()
        self.particle_encoder.model = _mlp3(input_dim, self.nf_particle, nf)
        self.relation_encoder = _Holder()
        self.relation_encoder.model = _mlp3(rel_input_dim, self.nf_relation, nf)
        self.particle_propagator = _Holder()
        self.particle_propagator.linear = nn.Linear(nf * 2, nf)
        self.relation_propagator = _Holder()
        self.relation_propagator.linear = nn.Linear(nf * 3, nf)
        self.non_rigid_predictor = _Holder()
        self.non_rigid_predictor.linear_0 = nn.Linear(nf, nf)
        self.non_rigid_predictor.linear_1 = nn.Linear(nf, nf)
        self.non_rigid_predictor.linear_2 = nn.Linear(nf, 3)
        for p in self.parameters():
            p.requires_grad_(False)
        self._engine = None
        self._uploaded_key = None
        self._precision = None       # None: the engine default (exact fp32, or AG_PRECISION)

    # ------------------------------------------------------------------ engine / weights
    def set_precision(self, mode):
        """'fp32': exact fp32 MFMA (default).  'bf16x3': 3-way bf16 split on the bf16 matrix pipe with fp32 accumulation
        - fp32-grade accuracy (same 1e-5 parity bar, tests/test_gpu_more.py), 1.32x faster end to end (BENCH_r02: 360.6 vs
        476.8 ms per 1024 x 20 cloth rollout)."""
        assert mode in ("fp32", "bf16x3")
        self._precision = mode
        if self._engine is not None:
            self._engine.set_precision(mode)

    def engine(self, device=None):
        dev = _require_gpu(device if device is not None else self.device)
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(dev, pstep=self.model_config["pstep"], n_his=self.n_his, rel_dim=self.rel_input_dim,
                                  motion_clamp=float(self.motion_clamp))
            self._uploaded_key = None
            if self._precision is not None:
                self._engine.set_precision(self._precision)
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key != self._uploaded_key:
            self._engine.load_state_dict_tensors(self.state_dict())
            self._uploaded_key = key
        return self._engine

    # ------------------------------------------------------------------ forward (model.py:130-342)
    @torch.no_grad()
    def forward(self, state, attrs, Rr=None, Rs=None, p_instance=None, action=None, particle_den=None, obj_mask=None,
                edges: EdgeList | None = None, **kwargs):
        dev = _require_gpu(state.device)
        eng = self.engine(dev)
        B, N = attrs.size(0), attrs.size(1)
        n_p, n_inst = p_instance.size(1), p_instance.size(2)
        n_s = N - n_p
        physics_keys = [k for k in kwargs.keys() if k.endswith("_physics_param")]
        assert len(physics_keys) == 1                                            # model.py:186-187
        pp = kwargs[physics_keys[0]].to(device=dev, dtype=torch.float32)
        if pp.size(-1) == 1:
            pp = pp[:, None, :].repeat(1, n_p, 1)                                # model.py:191-197
        else:
            pp = pp.reshape(B, n_p, 1)                                           # model.py:204
        phys = torch.cat([pp[..., 0], torch.zeros(B, n_s, device=dev)], 1).contiguous()   # model.py:206-207
        assert action is not None                                                # model.py:222
        group = torch.cat([p_instance.to(torch.float32), torch.zeros(B, n_s, n_inst, device=dev)], 1).contiguous()
        if edges is None:
            assert Rr is not None and Rs is not None
            edges = EdgeList.from_dense(Rr, Rs)
        assert edges.N == N
        state = state.to(torch.float32).contiguous()
        attrs = attrs.to(torch.float32).contiguous()
        action = action.to(torch.float32).contiguous()
        assert state.shape == (B, self.n_his, N, 3)
        pred_pos = torch.empty((B, n_p, 3), device=dev, dtype=torch.float32)
        pred_motion = torch.empty((B, n_p, 3), device=dev, dtype=torch.float32)
        eng.check(eng.lib.ag_forward(eng.ctx, current_stream(dev), ptr(state), ptr(attrs), ptr(action), ptr(phys),
                                     ptr(group), n_inst, ptr(edges.recv), ptr(edges.send), ptr(edges.row_ptr),
                                     ptr(edges.n_edges), edges.edge_cap, B, N, n_p, ptr(pred_pos), ptr(pred_motion)))
        return pred_pos, pred_motion
