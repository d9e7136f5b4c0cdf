"""Particle-swarm pre-tracker with the reference's interface (RandomOptimizer.py:10-227).

``RandomOptimizer(cfg, mipsfusion)`` and ``optimize(model, depth_img, initial_pose, last_frame_pose, n_iter)``
take and return what the reference's do.  The difference is where a round runs: the search state (rotation,
translation, search size) stays in a 32-float device buffer, a round is five kernel launches
(ro_particles -> hashgrid_fwd -> decoder_fwd -> ro_fitness -> ro_update) and nothing is read back until the last
round, whereas the reference issues ~25 eager ops and one device->host synchronisation (``if success_flag``) per
round.  The small helper methods keep their reference semantics for callers that use them directly.
"""
import numpy as np
import torch

from . import dist as mdist
from . import ops
from ._lib import FEAT_LEVEL_MAJOR
from .helper_functions.geometry_helper import quaternion_to_matrix
from .helper_functions.sampling_helper import sample_pixels_uniformly


# sample order of a round: point-major (hash-grid lookups 144 -> 77 us per round; see mipsf_ro_particles(point_major))
_POINT_MAJOR = True


class RandomOptimizer:
    def __init__(self, cfg, mipsfusion):
        self.cfg = cfg
        self.slam = mipsfusion
        self.dataset = self.slam.dataset
        self.device = self.slam.device
        ro = cfg["tracking"]["RO"]
        self.particle_size = ro["particle_size"]
        self.scaling_coefficient1 = ro["initial_scaling_factor"]
        self.scaling_coefficient2 = ro["rescaling_factor"]
        self.sdf_weight = 1000.
        self.trunc_value = cfg["training"]["trunc"]
        # extension: split the swarm over the ranks of the default process group (each holds a replica of the sub-map)
        self.particle_split = False
        # arithmetic of the decoder inside a round.  Default "bf16x6" (fp32 operands carried exactly as three bf16 pieces,
        # six products: the reference's fp32 arithmetic): the update compares fitness values (`fitness < fitness[0]`) and
        # weighs particles by differences of them, so the default reproduces the reference's pose to 1e-4
        # (tests/golden/ro.npz).  "f16x3" (hi/lo f16 operands, 22-23 bits, 1.6x faster rounds) does too.  "f16" (plain f16
        # operands, BASELINE config 5 "fp16 decoder on CDNA4", fastest) is an explicit opt-in: its 2e-3 fitness noise can
        # flip which particles count as advanced; the tracked pose stays within 1e-3 of the reference's
        # (test_random_optimizer_f16_rounds_track_the_reference_pose) -- callers that choose it state that tolerance.
        self.decoder_precision = "bf16x6"

        # particle swarm template, same draw as RandomOptimizer.py:26-33 (numpy global RNG)
        pst = np.random.multivariate_normal(np.zeros(6), np.eye(6), self.particle_size).astype(np.float32)
        pst = torch.from_numpy(pst)
        pst[0, :] = 0
        self.pre_sampled_particle = torch.clamp(pst, -2., 2.).to(self.device).contiguous()
        self.no_rel_trans = torch.tensor([1., 0., 0., 0., 0., 0., 0.]).to(self.device)

        self.iW = cfg["tracking"]["ignore_edge_W"]
        self.iH = cfg["tracking"]["ignore_edge_H"]
        self.rays_dir = self.dataset.rays_d
        self.row_indices, self.col_indices = sample_pixels_uniformly(self.dataset.H, self.dataset.W, ro["n_rows"],
                                                                     ro["n_cols"])
        self.fx, self.fy, self.cx, self.cy = self.dataset.fx, self.dataset.fy, self.dataset.cx, self.dataset.cy
        self.intrinsic = torch.tensor([[self.fx, 0., self.cx], [0., self.fy, self.cy], [0., 0., 1.]]).to(self.device)
        # the five lattice offsets of optimize() never change: their camera-frame directions live on the device
        rows, cols = self.row_indices, self.col_indices
        self._dirs = torch.stack([self.rays_dir[rows + o, cols + o, :] for o in range(5)]).to(
            self.device, torch.float32).contiguous()

    # ------------------------------------------------------------- reference helpers (same semantics, torch)
    def pose_6D_to_7D(self, batch_pose):
        s = batch_pose[:, 0] ** 2 + batch_pose[:, 1] ** 2 + batch_pose[:, 2] ** 2
        qw = torch.where(s <= 1., torch.sqrt(1 - s), torch.zeros_like(s)).unsqueeze(1)
        return torch.cat([qw, batch_pose], dim=-1)

    def get_abs_pose(self, ref_pose_rot, ref_pose_trans, particle_template):
        return ref_pose_rot @ quaternion_to_matrix(particle_template[:, :4]), \
            ref_pose_trans + particle_template[:, 4:, None]

    def batch_points_trans(self, points, pose_rot, pose_trans):
        return torch.transpose(pose_rot @ torch.transpose(points, 0, 1) + pose_trans, 1, 2)

    def get_fitness(self, model, abs_rot, abs_trans, last_frame_pose, target_d, rays_d_cam):
        world = self.batch_points_trans(rays_d_cam * target_d, abs_rot, abs_trans)
        mean_masked_sdf = ops.ro_fitness(model.run_network(world), target_d.reshape(-1).contiguous(), self.trunc_value)
        return mean_masked_sdf * self.sdf_weight, mean_masked_sdf

    def update_cur_pose(self, rot_cur, trans_cur, mean_transform):
        return rot_cur @ quaternion_to_matrix(mean_transform[:4]), trans_cur + mean_transform[4:][..., None]

    def update_search_size(self, mean_pred_sdf, mean_transform):
        s = torch.abs(mean_transform) + 0.0001
        return (self.scaling_coefficient2 * mean_pred_sdf * s / s.norm() + 0.0001)[None, ...]

    # ------------------------------------------------------------------------------------------ the fused loop
    def _enqueue_round(self, model, state, target_d, dirs, rc, packed, group=None):
        """One round (RandomOptimizer.py:177-224) on the current stream; no host synchronisation.

        With ``self.particle_split`` and an initialised process group the swarm is cut into contiguous shares
        (SURVEY 8e row 3): every rank holds a replica of the sub-map, evaluates its share's fitness, one all_gather
        (RCCL) reassembles [P, 8] = (mean masked |sdf|, 7-D pose) and every rank applies the identical update --
        fitness values are per-particle quantities, so the result equals the unsplit round bit for bit."""
        P_all, n = self.particle_size, dirs.shape[0]
        rank, world = mdist.rank_world(group) if self.particle_split else (0, 1)
        lo, hi = mdist.share_of(P_all, rank, world)
        P = hi - lo
        pst = self.pre_sampled_particle if world == 1 else self.pre_sampled_particle[lo:hi]
        # point-major sample order: a hash-grid wavefront = 64 particles' copies of one lattice point (same cells)
        xn, pst7 = ops.ro_particles(pst, state, dirs, target_d, rc, point_major=_POINT_MAJOR)
        feat = ops.hashgrid_fwd(xn, model.embed_fn.params.detach(), model.embed_fn.meta, FEAT_LEVEL_MAJOR)
        sdf = ops.decoder_fwd_sdf(packed if self.decoder_precision == "f32" else None, feat, FEAT_LEVEL_MAJOR, xn, None,
                                  P * n, precision=self.decoder_precision,       # SDF column only (scene_rep.py:106-107)
                                  packed16=None if self.decoder_precision == "f32" else packed)
        mean_masked = ops.ro_fitness(sdf.view(P, n, 1), target_d, self.trunc_value, point_major=_POINT_MAJOR)
        if world > 1:
            rows = mdist.gather_particle_results(torch.cat([mean_masked[:, None], pst7], 1), P_all, group)
            mean_masked, pst7 = rows[:, 0].contiguous(), rows[:, 1:].contiguous()
        ops.ro_update(mean_masked, pst7, state, self.sdf_weight, self.scaling_coefficient2)
        return mean_masked

    # ------------------------------------------------------------------------------ captured form of optimize()
    @torch.no_grad()
    def capture(self, model, n_iter, stream):
        """Record the n_iter rounds of ``optimize`` once as a hipGraph (5 launches per round + the operand-image pack);
        ``optimize_graphed`` then costs two small uploads, one replay and one 32-float read-back per frame instead of
        ~30 eager launches with their Python between them.  The graph reads the model's parameter tensors in place, so
        it stays valid while the map is optimised (not across ``load_state_dict`` into NEW tensors).  stream: the
        non-default work stream (mipsfusion_amd.graph.work_stream)."""
        dev = self.device
        n = self.row_indices.shape[0]
        self._g_state = torch.zeros(ops.RO_STATE_FLOATS, dtype=torch.float32, device=dev)
        self._g_td5 = torch.zeros(5, n, dtype=torch.float32, device=dev)
        self._g_state_host = torch.zeros(ops.RO_STATE_FLOATS, dtype=torch.float32).pin_memory()
        self._g_td5_host = torch.zeros(5, n, dtype=torch.float32).pin_memory()
        self._g_out_host = torch.zeros(ops.RO_STATE_FLOATS, dtype=torch.float32).pin_memory()
        rc = model._rc(1, 0)

        def rounds():
            ws = model.decoder.ordered_parameters()
            packed = ops.decoder_pack(ws) if self.decoder_precision == "f32" else ops.decoder_pack16(ws, precision=self.decoder_precision)
            for i in range(n_iter):
                o = i % 5
                self._enqueue_round(model, self._g_state, self._g_td5[o], self._dirs[o], rc, packed)
        with torch.cuda.stream(stream):
            rounds()                                            # allocator warm-up on the capture stream
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=stream):
                rounds()
        self._g_n_iter, self._g_model = n_iter, model
        self._lattice_flat = torch.stack([(self.row_indices + o) * self.dataset.W + (self.col_indices + o)
                                          for o in range(5)]).to(dev)

    @torch.no_grad()
    def enqueue_graphed(self, depth_flat_dev, initial_pose_cpu):
        """The frame's rounds, enqueued and NOT waited for: the tracked pose is left in words 0..11 of the device search
        state (``tracked_pose_dev``: 3x3 rotation row-major | translation) for a consumer on the same stream --
        ``ops.pose_handover`` hands it to the tracking iterations' pose Parameters without a host round trip."""
        h = self._g_state_host
        h.zero_()
        h[0:9] = initial_pose_cpu[:3, :3].reshape(9)
        h[9:12] = initial_pose_cpu[:3, 3]
        h[12:18] = float(self.scaling_coefficient1)
        self._g_state.copy_(h, non_blocking=True)
        torch.index_select(depth_flat_dev, 0, self._lattice_flat.reshape(-1), out=self._g_td5.view(-1))
        self._graph.replay()

    @property
    def tracked_pose_dev(self):
        return self._g_state

    def optimize_graphed(self, depth_flat_dev, initial_pose_cpu, waiting=None):
        """depth_flat_dev: the frame's depth as a flat DEVICE tensor [H*W] (the lattice is gathered on the device);
        initial_pose_cpu: [4,4] CPU tensor -> tracked pose [4,4] on the CPU.  Same arithmetic as ``optimize``.
        waiting: optional ``f(fn)`` that runs the blocking read-back ``fn`` (mipsfusion_amd.sequence: lets the sample
        producer threads use the host while this thread waits for the GPU)."""
        self.enqueue_graphed(depth_flat_dev, initial_pose_cpu)
        self._g_out_host.copy_(self._g_state, non_blocking=True)
        if waiting is None:
            torch.cuda.current_stream().synchronize()
        else:
            waiting(torch.cuda.current_stream().synchronize)
        pose = torch.eye(4, dtype=torch.float32)
        pose[:3, :3] = self._g_out_host[0:9].view(3, 3)
        pose[:3, 3] = self._g_out_host[9:12]
        return pose

    @torch.no_grad()
    def optimize(self, model, depth_img, initial_pose, last_frame_pose, n_iter=10, return_state=False):
        if n_iter <= 0:
            return initial_pose
        dev = self.device
        rows, cols = self.row_indices, self.col_indices
        # depth of the five lattice offsets: one small upload per frame instead of one per round
        td5 = torch.stack([depth_img[rows + o, cols + o] for o in range(5)]).to(dev, torch.float32).contiguous()
        state = torch.zeros(ops.RO_STATE_FLOATS, dtype=torch.float32, device=dev)
        init = initial_pose.to(dev, torch.float32)
        state[0:9] = init[:3, :3].reshape(9)
        state[9:12] = init[:3, 3]
        state[12:18] = float(self.scaling_coefficient1)
        rc = model._rc(1, 0)
        ws = model.decoder.ordered_parameters()
        packed = ops.decoder_pack(ws) if self.decoder_precision == "f32" else ops.decoder_pack16(ws, precision=self.decoder_precision)
        for i in range(n_iter):
            o = i % 5
            self._enqueue_round(model, state, td5[o], self._dirs[o], rc, packed)
        pose = torch.eye(4, dtype=torch.float32, device=dev)
        pose[:3, :3] = state[0:9].view(3, 3)
        pose[:3, 3] = state[9:12]
        return (pose, state) if return_state else pose
