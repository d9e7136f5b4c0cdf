"""``JointEncoding`` -- the drop-in boundary of the hot path (reference: model/scene_rep.py:11-238).

Same constructor, attributes (``embed_fn``, ``embedpos_fn``, ``decoder``), methods and state-dict keys as the
reference class, so ``mipsfusion.py`` / ``InactiveMap.py`` / ``RandomOptimizer.py`` / ``Mesher.py`` can import it
unchanged.  Internally the ~390 eager ops of one reference iteration become ten HIP kernels:

    sample_rays -> hashgrid_fwd -> decoder_fwd -> render_fwd (+ loss_finalize)
    render_bwd  -> decoder_bwd -> decoder_wgrad (+ reduce) -> hashgrid_bwd -> rays_bwd

chained through three ``torch.autograd.Function`` objects (placement, query, render/loss) so that gradients
reach the grid, the decoder and -- through ``rays_o`` / ``rays_d`` -- the pose parameters of the caller.
"""
import copy

import numpy as np
import os
import torch
import torch.nn as nn

from .. import ops
from .._lib import FEAT_LEVEL_MAJOR
from .decoder import MLP_reg
from .encodings import get_encoder


class _PlaceFn(torch.autograd.Function):
    """scene_rep.py:156-179 + :134-142: (rays, depth, noise) -> z_vals, normalised sample coordinates."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, target_d, noise, tables, rc, N, S):
        rays_o, rays_d = ops._f32c(rays_o), ops._f32c(rays_d)
        z_vals, xn, counts = ops.sample_rays(rays_o, rays_d, target_d, noise, tables, rc, N, S)
        ctx.rc, ctx.N, ctx.S = rc, N, S
        ctx.save_for_backward(z_vals)
        ctx.mark_non_differentiable(z_vals, counts)
        ctx.set_materialize_grads(False)      # no zero tensors for the outputs nobody differentiates
        return z_vals, xn, counts

    @staticmethod
    def backward(ctx, _dz, dxn, _dc):
        if dxn is None:
            return (None,) * 8
        (z_vals,) = ctx.saved_tensors
        d_o, d_d = ops.rays_bwd(ops._f32c(dxn), z_vals, ctx.rc, ctx.N, ctx.S)
        return d_o, d_d, None, None, None, None, None, None


class _NormaliseFn(torch.autograd.Function):
    """run_network's normalisation (scene_rep.py:138-142), float64 inside the kernel."""

    @staticmethod
    def forward(ctx, pts, rc):
        ctx.rc = rc
        return ops.normalise_points(ops._f32c(pts), rc)

    @staticmethod
    def backward(ctx, dxn):
        return ops.normalise_bwd(ops._f32c(dxn), ctx.rc), None


# below this many samples the routing kernels take a few microseconds and a second stream only adds launch work
_ROUTE_AHEAD_MIN_M = 65536
# forward-only queries of 2^24 samples or more are cut into launches of 2^23 (see JointEncoding._query)
_MAX_QUERY, _QUERY_CHUNK = 1 << 24, 1 << 23


class _QueryFn(torch.autograd.Function):
    """query_color_sdf (scene_rep.py:118-128) fused: hash grid (level-major features) -> decoder with the
    frequency encoding computed in its prologue.

    Parameter gradients are returned through autograd by default (``torch.autograd.grad``, hooks, retain_graph and
    partial backwards all behave as for any torch module).  ``JointEncoding.accumulate_param_grads_in_place = True``
    (opt-in, used by bench.py and the captured training loops) makes the backward add them straight into ``.grad``
    instead -- no 36 MB zero-fill + add pass per iteration -- which is only valid for plain ``loss.backward()``
    accumulation loops; it is ignored for a parameter that is not a leaf or carries tensor hooks."""

    @staticmethod
    def forward(ctx, xn, owner, grid_params, *weights):
        xn = ops._f32c(xn)
        M = xn.shape[0]
        meta = owner.embed_fn.meta
        # when the points need a gradient (pose optimisation) the forward also keeps d feat / d x, so the backward
        # never gathers the table a second time
        jac = None
        # the routing half of the grid's parameter-gradient scatter only needs the points: it runs on a second stream
        # next to this forward pass (joined before the forward returns, so nothing is left dangling if no backward follows)
        routed = None
        if ctx.needs_input_grad[2] and owner.route_ahead and M >= _ROUTE_AHEAD_MIN_M:
            routed = ops.hashgrid_route_ahead(xn, meta)
        if ctx.needs_input_grad[0]:
            feat, jac = ops.hashgrid_fwd(xn, grid_params.detach(), meta, FEAT_LEVEL_MAJOR, with_jac=True)
        else:
            feat = ops.hashgrid_fwd(xn, grid_params.detach(), meta, FEAT_LEVEL_MAJOR)
        need = any(ctx.needs_input_grad)
        prec = owner.decoder_precision
        if prec == "f16" and need:
            raise RuntimeError('decoder_precision "f16" is forward-only (use "bf16x6", "f16x3" or "f32" when gradients are needed)')
        # one operand-image buffer per arithmetic: `packed` (fp32 images) or `packed16` (f16 hi/lo images, forward and
        # backward chain); it is saved for the backward under the same name
        cache = owner._frozen_pack                  # see JointEncoding.frozen_weights(): the map does not change in here
        if cache is not None and prec in cache:
            packed, packed16 = cache[prec]
        else:
            if prec == "f32":
                packed, packed16 = ops.decoder_pack(weights), None
            else:
                packed, packed16 = None, ops.decoder_pack16(weights, precision=prec)
            if cache is not None:
                cache[prec] = (packed, packed16)
        # lean record: when the weight gradients will come from the streaming f16 kernel (the default behind the f16x3
        # chain) H1 is recomputed there from x and the forward does not write it (a third of the record)
        lean = bool(need and prec in ops.SPLIT_PRECISIONS and owner.wgrad_precision in ("auto", "stream_" + prec) and owner.lean_record)
        # a frozen decoder (tracking: only the points need a gradient): the chain reads the ReLU masks and nothing else of the
        # record -- the 1 KB of activations per sample would be written for nobody
        masks_only = bool(need and prec in ops.SPLIT_PRECISIONS and not any(ctx.needs_input_grad[3:]))
        out, saved = ops.decoder_fwd(packed, feat, FEAT_LEVEL_MAJOR, xn, None, M,
                                     save="masks" if masks_only else ("lean" if lean else need), precision=prec, packed16=packed16)
        ctx.lean = lean and not masks_only
        ctx.tile_live = getattr(saved, "mipsf_tile_live", None)    # (python attributes do not travel with saved tensors)
        if packed is None:
            packed = packed16
        ctx.prec = prec
        if routed is not None:
            torch.cuda.current_stream(xn.device).wait_event(routed[1])
        ctx.routed = routed
        ctx.owner, ctx.M, ctx.meta, ctx.has_jac = owner, M, meta, jac is not None
        ctx.save_for_backward(xn, feat, out, saved, packed, grid_params, *weights, *([jac] if jac is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        xn, feat, out, saved, packed, grid_params, *weights = ctx.saved_tensors
        jac = weights.pop() if ctx.has_jac else None
        need_w = any(ctx.needs_input_grad[3:])
        need_g = ctx.needs_input_grad[2]
        need_x = ctx.needs_input_grad[0]
        if ctx.tile_live is not None:
            saved.mipsf_tile_live = ctx.tile_live
        direct = ctx.owner.accumulate_param_grads_in_place and all(
            p.is_leaf and not p._backward_hooks for p in (grid_params, *weights))
        # frozen parameters (requires_grad False, e.g. the map during tracking) skip their kernels entirely
        grads = None
        if need_w:
            grads = []
            for w, need in zip(weights, ctx.needs_input_grad[3:]):
                if direct and need:
                    if w.grad is None:
                        w.grad = torch.zeros_like(w)
                    grads.append(w.grad)
                else:
                    grads.append(torch.zeros_like(w))
        dfeat, dx, _, tiles = ops.decoder_bwd(packed, feat, FEAT_LEVEL_MAJOR, xn, None, out, ops._f32c(dout), saved, grads,
                                              ctx.M, precision=ctx.prec, packed16=packed if ctx.prec != "f32" else None,
                                              wgrad_precision=("stream_" + ctx.prec) if ctx.lean else ctx.owner.wgrad_precision,
                                              recompute_h1=ctx.lean, return_tiles=True)
        dparams = None
        fresh_grad = False
        if need_g:
            if direct:
                if grid_params.grad is None:
                    grid_params.grad = torch.zeros_like(grid_params)
                    fresh_grad = True
                dparams = grid_params.grad
            else:
                dparams = torch.zeros_like(grid_params)
        if need_g:
            # (a fresh zeros tensor, or the caller's word that the optimiser left .grad zero: slices are stored, not added)
            ops.hashgrid_bwd(xn, grid_params.detach(), dfeat, dparams, ctx.meta, FEAT_LEVEL_MAJOR, None, routed=ctx.routed,
                             dparams_zero=(not direct) or fresh_grad or ctx.owner.grid_grad_is_zero_at_backward)
            ctx.routed = None
        if need_x:
            ops.hashgrid_dx_from_jac(jac, dfeat, dx, ctx.meta, FEAT_LEVEL_MAJOR, tiles=tiles)   # (dfeat is untouched since the chain wrote it)
        w_out = [None] * len(weights)
        if need_w and not direct:
            w_out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[3:])]
        return (dx if need_x else None, None, None if direct else dparams, *w_out)


class _RenderFn(torch.autograd.Function):
    """raw2outputs / sdf2weights (+ the four training losses): scene_rep.py:58-103, 211-236."""
    fuse_backward = os.environ.get("MIPSF_RENDER_FUSE_BWD", "1") != "0"    # the objective's gradient in the forward launch

    @staticmethod
    def forward(ctx, raw, z_vals, target_rgb, target_d, counts, rc, N, S, train, loss_w=None, share_of=None):
        raw = ops._f32c(raw)
        total = None
        ctx.n_norm = None
        ctx.draw, ctx.draw_stale = None, False
        if train and share_of is not None:   # a share of a ray-data-parallel batch: the losses of the WHOLE batch
            rgb, depth, var, disp, acc, _, losses, total, ctx.n_norm = ops.render_fwd(
                raw, z_vals, target_rgb, target_d, counts, rc, N, S, train, loss_weights=loss_w, share_of=share_of)
        elif train and loss_w is not None:     # the weighted objective comes out of the loss kernel itself
            if ctx.needs_input_grad[0] and S <= ops.RENDER_DRAW_MAX_S and _RenderFn.fuse_backward:
                # ... and its gradient with it (for an objective gradient of exactly 1: `loss.backward()`)
                rgb, depth, var, disp, acc, _, losses, total, ctx.draw = ops.render_fwd(
                    raw, z_vals, target_rgb, target_d, counts, rc, N, S, train, loss_weights=loss_w, want_draw=True)
            else:
                rgb, depth, var, disp, acc, _, losses, total = ops.render_fwd(raw, z_vals, target_rgb, target_d, counts, rc, N,
                                                                              S, train, loss_weights=loss_w)
        else:
            rgb, depth, var, disp, acc, _, losses = ops.render_fwd(raw, z_vals, target_rgb, target_d, counts, rc, N, S,
                                                                   train)
        ctx.rc, ctx.N, ctx.S, ctx.train = rc, N, S, train
        ctx.save_for_backward(raw, z_vals, target_rgb, target_d, counts, losses, loss_w if total is not None else None)
        ctx.mark_non_differentiable(var, disp, acc)
        ctx.set_materialize_grads(False)      # rgb / depth usually carry no gradient in training: skip their zero fills
        if total is not None:
            return rgb, depth, var, disp, acc, losses, total.reshape(())
        if train:
            return rgb, depth, var, disp, acc, losses
        return rgb, depth, var, disp, acc

    @staticmethod
    def backward(ctx, g_rgb, g_depth, _gv, _gd, _ga, g_losses=None, g_total=None):
        raw, z_vals, target_rgb, target_d, counts, losses, loss_w = ctx.saved_tensors
        g_rgb = ops._f32c(g_rgb) if g_rgb is not None else None
        g_depth = ops._f32c(g_depth) if g_depth is not None else None
        g_total = ops._f32c(g_total).reshape(1) if (g_total is not None and loss_w is not None) else None
        if ctx.draw is not None and not ctx.draw_stale and g_total is not None and g_losses is None and g_rgb is None and g_depth is None:
            # the forward launch has written d objective / d raw for an objective gradient of exactly 1
            one = ops.unit_grad(raw.device)
            if g_total.data_ptr() == one.data_ptr() and one._version == 0:      # (never written in place since torch.ones made it)
                return ctx.draw, None, None, None, None, None, None, None, None, None, None     # ... and this IS that gradient
            ctx.draw_stale = True         # the kernel decides on the device (g_total == 1: returns at once), so after this call
            draw = ops.render_bwd(raw, z_vals, target_rgb, target_d, counts, losses, ctx.rc, None, None, None, ctx.N, ctx.S,
                                  g_total=g_total, loss_weights=loss_w, keep_draw=ctx.draw)     # the buffer may hold either
            return draw, None, None, None, None, None, None, None, None, None, None
        if ctx.train and g_losses is None and g_total is None:
            g_losses = torch.zeros(8, dtype=torch.float32, device=raw.device)
        draw = ops.render_bwd(raw, z_vals, target_rgb, target_d, counts, losses, ctx.rc,
                              ops._f32c(g_losses) if (ctx.train and g_losses is not None) else None, g_rgb, g_depth,
                              ctx.N, ctx.S, g_total=g_total, loss_weights=loss_w, n_norm=ctx.n_norm)
        return draw, None, None, None, None, None, None, None, None, None, None


class JointEncoding(nn.Module):
    def __init__(self, config, bound_box, coords_norm_factor):
        super().__init__()
        self.config = config
        self.bounding_box = bound_box
        self.coords_norm_factor = coords_norm_factor
        self._bound64 = torch.as_tensor(bound_box).detach().to("cpu", torch.float64).tolist()
        self._half64 = torch.as_tensor(coords_norm_factor).detach().to("cpu", torch.float64).reshape(-1).tolist()
        # extension, off by default: see _QueryFn (plain `loss.backward()` loops may opt in)
        self.accumulate_param_grads_in_place = False
        # with accumulate_param_grads_in_place: the caller's word that the table's .grad is all zero whenever a backward
        # pass starts (ONE backward per optimiser step, and the optimiser clears the gradients: FusedAdam.step(zero_grad=True)
        # / the captured loops) -- the grid scatter then stores its slices instead of adding to the old values.  Wrong
        # gradients if the promise is broken (several backward passes accumulated before a step): off by default.
        self.grid_grad_is_zero_at_backward = False
        # arithmetic of the decoder (csrc/decoder16.hip, csrc/decoder.hip):
        #   "bf16x6" (default) every fp32 operand -- weight and activation -- carried EXACTLY as three bf16 pieces, a product =
        #           six MFMAs on the bf16 matrix cores, fp32 accumulate: the arithmetic of the reference's fp32 nn.Linear
        #           layers (model/decoder.py:32-50); passes the reference goldens at the fp32 kernels' own gates and sits
        #           where they sit against fp64 truth, at 1.6x their speed
        #   "f16x3" the fast mode: f16 matrix cores on hi/lo split operands (22-23 significant bits), three MFMAs per
        #           product, fp32 accumulate: ~3e-7 relative (2-3x the fp32 error), 1.5x faster than "bf16x6" again
        #   "f32"   fp32-input matrix cores: exact fp32 products (the round-1 path)
        #   "f16"   plain f16 operands, forward-only (2e-3 of the output range): set by consumers that state a tolerance
        self.decoder_precision = "bf16x6"
        # weight-gradient kernel (ops.decoder_bwd): "auto" = the streaming f16 hi/lo kernel behind the f16x3 chain
        # (fp32-class, read-bandwidth bound), the fp32 LDS kernel otherwise; "f32" / "stream_bf16x6" / ... force one.
        # (The ~5e-6 arithmetics "bf16x3" / "stream_bf16x3" pass every per-step tolerance, but the chaotic 51-iteration
        # sequence drifts 10x further from the reference's run with them.)
        self.wgrad_precision = "auto"
        # with the streaming f16 weight-gradient kernel: keep the lean activation record (no H1) and recompute H1 there
        self.lean_record = True
        # opt-in: run the routing half of the hash grid's backward on a second stream next to the forward pass
        # (ops.hashgrid_route_ahead).  Off by default: measured on the headline workload the routing kernels, squeezed in
        # beside the persistent decoder forward, take 211 us instead of 70 and slow that kernel from 99 to 137 us -- the
        # step gets 40 us LONGER (0.88 -> 0.92 ms).  It pays only where the forward leaves CUs idle.
        self.route_ahead = False
        # ray-data-parallel training (mipsfusion_amd/ray_dp.py sets it): the rays handed to forward() are ONE SHARE of the
        # iteration's batch; the callable sums a small fp64 vector over the ranks, the losses (and their gradients) are then
        # those of the WHOLE batch -- fs_weight / sdf_weight from the batch's counts (helper_functions/utils.py:43-47), the
        # depth loss over the batch's valid rays (scene_rep.py:218), every mean over all N rays -- on every rank
        self.ray_share_reduce = None
        self._frozen_pack = None
        self._tables = {}
        self._jitter_ring = {}
        self.get_resolution()
        self.get_encoding(config)
        self.get_decoder(config)
        self.save_initial_param()

    # ------------------------------------------------------------------ construction (scene_rep.py:22-55)
    def get_resolution(self):
        dim_max = max(b[1] - b[0] for b in self._bound64)
        if self.config["grid"]["voxel_sdf"] > 10:
            self.resolution_sdf = self.config["grid"]["voxel_sdf"]
        else:
            self.resolution_sdf = int(dim_max / self.config["grid"]["voxel_sdf"])

    def get_encoding(self, config):
        self.embedpos_fn, self.input_ch_pos = get_encoder(config["pos"]["enc"], n_bins=config["pos"]["n_bins"])
        self.embed_fn, self.input_ch = get_encoder(config["grid"]["enc"],
                                                   log2_hashmap_size=config["grid"]["hash_size"],
                                                   desired_resolution=256)
        if self.embed_fn.otype != "hashgrid" or self.embedpos_fn.otype != "frequency" or self.input_ch_pos != 48:
            raise ValueError("the fused HIP path is built for grid.enc=HashGrid + pos.enc=Frequency(n_bins=8), "
                             "the only combination the reference's configs use")

    def get_decoder(self, config):
        self.decoder = MLP_reg(config, input_ch=self.input_ch, input_ch_pos=self.input_ch_pos)

    def save_initial_param(self):
        self.initial_dict = copy.deepcopy(self.state_dict())

    def recover_initial_param(self):
        self.load_state_dict(self.initial_dict)

    # --------------------------------------------------------------------------------- helpers
    def _rc(self, n_uniform, n_near, emd_w=0.0):
        return ops.make_render_cfg(self.config, self._bound64, self._half64, n_uniform, n_near, emd_w)

    def _linspace_tables(self, device, guided: bool):
        key = (str(device), guided)
        if key not in self._tables:
            tr, cam = self.config["training"], self.config["cam"]
            if guided:
                zu = torch.linspace(cam["near"], cam["far"], max(0, tr["n_samples_d"]))
                zoff = torch.linspace(-tr["range_d"], tr["range_d"], steps=tr["n_range_d"])
                znd = torch.linspace(cam["near"], cam["far"], steps=tr["n_range_d"])
                self._tables[key] = tuple(t.to(device=device, dtype=torch.float32).contiguous() for t in (zu, zoff, znd))
            else:
                zu = torch.linspace(cam["near"], cam["far"], tr["n_samples"]).to(device=device, dtype=torch.float32)
                self._tables[key] = (zu.contiguous(), None, None)
        return self._tables[key]

    def _query(self, x32):
        # the 16-bit decoder kernels address a call's samples through 4 GB buffer resources: at most 2^24 - 1 samples per
        # launch.  Forward-only consumers (a 256^3 mesher grid in one call, Mesher.py:342-630) are cut into launches of 2^23
        M = x32.shape[0]
        if M >= _MAX_QUERY and not (torch.is_grad_enabled() and (x32.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return torch.cat([self._query(x32[i:i + _QUERY_CHUNK]) for i in range(0, M, _QUERY_CHUNK)], 0)
        return _QueryFn.apply(x32, self, self.embed_fn.params, *self.decoder.ordered_parameters())

    def __getstate__(self):
        d = self.__dict__.copy()        # (pinned staging buffers and their events stay with this process)
        d["_jitter_ring"], d["_frozen_pack"], d["ray_share_reduce"] = {}, None, None
        return d

    def __deepcopy__(self, memo):
        # a copy must not advance the CPU RNG (the constructor's nn.Linear initialisation would): the reference's
        # copy.deepcopy(model) (InactiveMap.py:67,81,107) draws nothing, and every later pixel-sampling call depends
        # on the generator state
        with torch.random.fork_rng(devices=[]):
            new = JointEncoding(self.config, self.bounding_box, self.coords_norm_factor)
        memo[id(self)] = new
        dev = self.embed_fn.params.device
        new.to(dev)
        new.load_state_dict(self.state_dict())
        new.initial_dict = copy.deepcopy(self.initial_dict)
        new.accumulate_param_grads_in_place = self.accumulate_param_grads_in_place
        new.grid_grad_is_zero_at_backward = self.grid_grad_is_zero_at_backward
        new.decoder_precision = self.decoder_precision
        new.wgrad_precision = self.wgrad_precision
        new.lean_record = self.lean_record
        new.route_ahead = self.route_ahead
        new._frozen_pack = None
        new.train(self.training)
        return new

    # ---------------------------------------------------------------- queries (scene_rep.py:105-146)
    def query_sdf(self, query_points):
        """scene_rep.py:105-108.  Without autograd (mesher grids, fitness probes) only the SDF branch of the decoder
        runs (the `[..., 3:4]` slice of scene_rep.py:106-107 computed alone): same values as column 3 of query_color_sdf, bit for bit."""
        needs_grad = torch.is_grad_enabled() and (query_points.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            return self.query_color_sdf(query_points)[..., 3:4]
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]]) / self.config["training"]["norm_factor"]
        if not flat.is_cuda:
            raise RuntimeError("JointEncoding runs on the GPU only (no CPU fallback)")
        xn = ops._f32c(flat)
        prec = self.decoder_precision
        ws = self.decoder.ordered_parameters()
        packed = ops.decoder_pack(ws) if prec == "f32" else None
        packed16 = ops.decoder_pack16(ws, precision=prec) if prec != "f32" else None
        out = []
        for i in range(0, max(1, xn.shape[0]), _QUERY_CHUNK if xn.shape[0] >= _MAX_QUERY else max(1, xn.shape[0])):
            xc = xn[i:i + _QUERY_CHUNK] if xn.shape[0] >= _MAX_QUERY else xn
            feat = ops.hashgrid_fwd(xc, self.embed_fn.params.detach(), self.embed_fn.meta, FEAT_LEVEL_MAJOR)
            out.append(ops.decoder_fwd_sdf(packed, feat, FEAT_LEVEL_MAJOR, xc, None, xc.shape[0], precision=prec, packed16=packed16))
        return (out[0] if len(out) == 1 else torch.cat(out, 0))[:, None]

    def query_color(self, query_points):
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def query_sdf_entropy_prob(self, query_points):
        return self.query_color_sdf(query_points)[..., 3:]

    def query_color_sdf(self, query_points):
        """query_points: ALREADY normalised coordinates [..., 3] -> [prod(...), 10]."""
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]]) / self.config["training"]["norm_factor"]
        if not flat.is_cuda:
            raise RuntimeError("JointEncoding runs on the GPU only (no CPU fallback)")
        return self._query(flat.to(torch.float32))

    def run_network(self, inputs):
        """inputs: UN-normalised local coordinates [..., 3] -> [..., 10]."""
        flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        if not flat.is_cuda:
            raise RuntimeError("JointEncoding runs on the GPU only (no CPU fallback)")
        rc = self._rc(1, 0)
        # the kernel applies /norm_factor itself (query_color_sdf's division, scene_rep.py:119)
        xn = _NormaliseFn.apply(flat, rc)
        out = self._query(xn)
        return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])

    # -------------------------------------------------------------- rendering (scene_rep.py:153-187)
    def _render(self, rays_o, rays_d, target_rgb, target_d, noise, train, emd_w):
        if not rays_o.is_cuda:
            raise RuntimeError("JointEncoding runs on the GPU only (no CPU fallback)")
        tr = self.config["training"]
        N = rays_o.shape[0]
        guided = target_d is not None
        # (training.n_samples_d = 0 is the reference's `z_vals = z_samples` branch, scene_rep.py:166-167: the merge with an empty
        # uniform list leaves the depth-guided samples in their own order)
        n_uniform = max(0, tr["n_samples_d"]) if guided else tr["n_samples"]
        n_near = tr["n_range_d"] if guided else 0
        S = n_uniform + n_near
        rc = self._rc(n_uniform, n_near, emd_w)
        if rc.perturb:
            if noise is None:
                # the reference draws this on the CPU default generator (scene_rep.py:176); doing the same keeps
                # the CPU RNG stream -- and therefore every later pixel-sampling call -- bit-identical
                noise = self._draw_jitter(N, S, rays_o.device)
            noise = ops._f32c(noise)
        else:
            noise = None
        tables = self._linspace_tables(rays_o.device, guided)
        td = ops._f32c(target_d).reshape(N, 1) if guided else None
        z_vals, xn, counts = _PlaceFn.apply(rays_o, rays_d, td, noise, tables, rc, N, S)
        raw = self._query(xn)
        trgb = ops._f32c(target_rgb) if train else None
        res = _RenderFn.apply(raw, z_vals, trgb, td if train else None, counts if train else None, rc, N, S, train,
                              self._objective_weights(raw.device) if train else None, self.ray_share_reduce if train else None)
        return res, z_vals, raw.reshape(N, S, 10)

    def _draw_jitter(self, N, S, device):
        """``torch.rand(N, S)`` of the default CPU generator, on the device.  The draw itself is made by the C replica of
        torch's generator (mipsfusion_amd/hostrng.py: bit-identical, checked against torch.rand at start-up, torch's own
        routine otherwise; 0.12 ms instead of 0.5-1.6 ms for 262 144 values) straight into one of four pinned buffers and goes
        up as an asynchronous copy -- the caller's thread does not wait for a pageable-memory staging copy."""
        from .. import hostrng
        if torch.cuda.is_current_stream_capturing():     # (a captured loop passes its jitter in: this draw would be baked into the graph)
            raise RuntimeError("JointEncoding drew its sample jitter on the host while a hipGraph was being captured: pass "
                               "`noise` (a static device tensor refilled before every replay) to forward / render_rays")
        n = N * S
        ring = self._jitter_ring.get(str(device))
        if ring is None or ring["cap"] < n:
            # ONE ring of four pinned buffers per device, sized for the largest batch seen so far and sliced per call (a
            # ring per (N, S) pinned 4 x N x S x 4 bytes for every distinct batch size a caller ever used)
            ring = self._jitter_ring[str(device)] = {"bufs": [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(4)],
                                                     "events": [None] * 4, "i": 0, "cap": n}
        i = ring["i"]
        ring["i"] = (i + 1) % 4
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()          # the upload that last read this buffer (four draws ago) is done
        buf = ring["bufs"][i][:n].view(N, S)
        with hostrng.session() as sess:
            sess.rand_(buf)
        out = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["events"][i] = ev
        return out

    def frozen_weights(self, begin: bool):
        """``frozen_weights(True)`` ... ``frozen_weights(False)`` brackets a stretch in which the decoder's weights do not
        change (the tracking iterations of a frame: pose-only optimisation, mipsfusion.py:456-575): the operand images
        are packed by the first forward inside and reused by the others (a 5 us launch per iteration otherwise; inside a
        captured tracking graph the pack is then recorded once per replay)."""
        self._frozen_pack = {} if begin else None

    def _objective_weights(self, device):
        """(rgb, depth, sdf, fs) weights of config["training"] as a device tensor, or None when the config has none: the
        loss kernel then also forms the objective of MIPSFusion.get_loss_from_ret (helper_functions.utils)."""
        tr = self.config.get("training", {})
        if not all(k in tr for k in ("rgb_weight", "depth_weight", "sdf_weight", "fs_weight")):
            return None
        vals = (float(tr["rgb_weight"]), float(tr["depth_weight"]), float(tr["sdf_weight"]), float(tr["fs_weight"]))
        key = (str(device), vals)
        if getattr(self, "_obj_w_key", None) != key:
            self._obj_w, self._obj_w_key = torch.tensor(vals, dtype=torch.float32, device=device), key
        return self._obj_w

    def forward_from_table(self, table, rows, rot, trans, fixed_poses, owner, noise, EMD_w=0.01, accumulate_in_place=False):
        """(extension, opt-in) The training branch of ``forward`` for rays that are ROWS of a device-resident ray table
        ``[..., 7]`` = (direction in the camera frame | rgb | depth) seen from poses ``cat([fixed_poses,
        qt_to_transform_matrix(rot, trans)])[owner]``: what ``ops.gather_pose_rays(...)`` followed by ``forward(rays_o,
        rays_d, rgb, depth, noise=noise)`` computes (keyframeSet.py:264-290, mipsfusion.py:320-322, scene_rep.py:156-238),
        with the row gather, the ray construction and the sample placement in ONE launch, and the ray + pose gradients of
        the backward in another (two launches fewer per iteration).  noise: [N, S] uniforms (required: the rays never
        exist on the host).  accumulate_in_place: as for ``ops.pose_rays``."""
        if not self.training:
            raise RuntimeError("forward_from_table is the training branch (use forward / render_rays for evaluation)")
        tr = self.config["training"]
        n_uniform, n_near = max(0, tr["n_samples_d"]), tr["n_range_d"]      # (0: scene_rep.py:166-167)
        S, N = n_uniform + n_near, rows.shape[0]
        rc = self._rc(n_uniform, n_near, float(EMD_w))
        if rc.perturb and noise is None:
            raise ValueError("forward_from_table needs the jitter noise as a device tensor")
        tables = self._linspace_tables(table.device, True)
        trgb, td, z_vals, xn, counts = ops.GatherPosePlaceFn.apply(rot, trans, fixed_poses, owner, table, rows,
                                                                   ops._f32c(noise) if rc.perturb else None, tables, rc, S,
                                                                   accumulate_in_place)
        raw = self._query(xn)
        res = _RenderFn.apply(raw, z_vals, trgb, td, counts, rc, N, S, True, self._objective_weights(raw.device),
                              self.ray_share_reduce)
        rgb, depth, losses = res[0], res[1], res[5]
        ret = {"rgb": rgb, "depth": depth, "rgb_loss": losses[0], "depth_loss": losses[1], "sdf_loss": losses[2],
               "fs_loss": losses[3], "psnr": losses[4:5].detach(), "_loss_vec": losses}
        if len(res) > 6:
            ret["_loss_total"], ret["_loss_total_weights"] = res[6], self._obj_w_key[1]
        return ret

    def render_rays(self, rays_o, rays_d, target_d=None, noise=None):
        (rgb, depth, var, disp, acc), z_vals, raw = self._render(rays_o, rays_d, None, target_d, noise, False, 0.0)
        return {"rgb": rgb, "depth": depth, "disp_map": disp, "acc_map": acc, "depth_var": var, "z_vals": z_vals,
                "raw": raw}

    def forward(self, rays_o, rays_d, target_rgb, target_d, EMD_w=0.01, noise=None):
        """Same contract as scene_rep.py:190-238.  ``noise`` (extension): an [N,S] U[0,1) tensor to use instead of
        drawing torch.rand on the CPU."""
        if not self.training:
            return self.render_rays(rays_o, rays_d, target_d=target_d, noise=noise)
        res, _z, _raw = self._render(rays_o, rays_d, target_rgb, target_d, noise, True, float(EMD_w))
        rgb, depth, losses = res[0], res[1], res[5]
        total = res[6] if len(res) > 6 else None
        # "_loss_vec" (extension): the kernel's loss vector itself, so that get_loss_from_ret can form the weighted
        # sum with one dot product instead of 4 selects + 4 scalings + 3 adds and their ~25 backward launches
        ret = {"rgb": rgb, "depth": depth, "rgb_loss": losses[0], "depth_loss": losses[1], "sdf_loss": losses[2],
               "fs_loss": losses[3], "psnr": losses[4:5].detach(), "_loss_vec": losses}
        if total is not None:   # "_loss_total" (extension): the objective with this model's own config weights, from the kernel
            ret["_loss_total"], ret["_loss_total_weights"] = total, self._obj_w_key[1]
        return ret
