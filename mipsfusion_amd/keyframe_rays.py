"""Device-resident keyframe ray database.

The reference keeps ``KeyframeSet.rays [num_kf, rays_per_kf, 7]`` on the CPU (model/keyframeSet.py:25), gathers the
sampled rows on the CPU every mapping iteration and uploads them in three slices (mipsfusion.py:296-317).  Here the
database lives in HBM; only the *indices* are generated on the host -- with the same ``random.sample`` calls in the
same order, so the index stream (and therefore every sampled ray) is bit-identical to the reference's for the same
seed -- and one gather kernel produces the rows on the device.

The sampling methods keep the reference's names, arguments and return values (rays on the device instead of the
CPU); everything else of ``KeyframeSet`` (sub-map bookkeeping, overlap masks) is host control plane and stays where
it is.
"""
import random

import torch

from . import ops


class DeviceRayDB:
    def __init__(self, num_kf: int, num_rays_to_save: int, device, storage: torch.Tensor = None):
        """storage: optional caller-owned device buffer of >= num_kf*num_rays_to_save*7 floats to keep the rows in (lets
        a caller put the database and the current frame's rays in ONE table that a captured iteration gathers from)."""
        self.num_rays_to_save = num_rays_to_save
        self.device = torch.device(device)
        if storage is None:
            self.rays = torch.zeros((num_kf, num_rays_to_save, 7), dtype=torch.float32, device=self.device)
        else:
            if storage.dtype != torch.float32 or not storage.is_contiguous() or storage.device.type != self.device.type:
                raise ValueError("storage must be a contiguous float32 tensor on the database's device")
            self.rays = storage.reshape(-1)[:num_kf * num_rays_to_save * 7].view(num_kf, num_rays_to_save, 7)

    # keyframeSet.py:170-175 -- the down-sampled rays of a new keyframe go straight to HBM (one 7 x R x 4 B upload)
    def store(self, kf_index: int, rays) -> None:
        self.rays[kf_index].copy_(rays.reshape(self.num_rays_to_save, 7), non_blocking=True)

    def _gather(self, flat_idx: torch.Tensor):
        n_rows = self.rays.shape[0] * self.rays.shape[1]
        if flat_idx.numel() and not flat_idx.is_cuda:      # host index stream: check before it becomes an HBM address
            lo, hi = int(flat_idx.min()), int(flat_idx.max())
            if lo < -n_rows or hi >= n_rows:
                raise IndexError(f"ray index out of range: [{lo}, {hi}] for a database of {n_rows} rows")
        return ops.gather_rays(self.rays, flat_idx.to(self.device, torch.int64, non_blocking=True))

    # ------------------------------------------------------------------ keyframeSet.py:268-275
    def sample_global_rays(self, bs: int, num_kf: int):
        idxs = torch.tensor(random.sample(range(num_kf * self.num_rays_to_save), bs))
        kf_ids = torch.div(idxs, self.num_rays_to_save, rounding_mode="floor")
        return self._gather(idxs), kf_ids

    # ------------------------------------------------------------------ keyframeSet.py:284-290
    def sample_rays_from_given(self, kf_Ids, bs: int):
        num_kf = kf_Ids.shape[0]
        idxs = torch.tensor(random.sample(range(num_kf * self.num_rays_to_save), bs))
        kf_indices = torch.div(idxs, self.num_rays_to_save, rounding_mode="floor")
        flat = kf_Ids.to(torch.int64)[kf_indices] * self.num_rays_to_save + (idxs - kf_indices * self.num_rays_to_save)
        return self._gather(flat), kf_indices

    # ------------------------------------------------------------------ keyframeSet.py:445-455
    def sample_rays_in_given_kf(self, given_kf_ids, pix_num: int):
        n = given_kf_ids.shape[0]
        idx = torch.tensor(random.sample(range(n * self.num_rays_to_save), pix_num))
        kf_indices = torch.div(idx, self.num_rays_to_save, rounding_mode="floor")
        kf_ids = given_kf_ids[kf_indices]
        flat = kf_ids.to(torch.int64) * self.num_rays_to_save + (idx - kf_indices * self.num_rays_to_save)
        return self._gather(flat), kf_ids, kf_indices

    # ------------------------------------------------------------------ keyframeSet.py:386-436
    def sample_rays_in_submap(self, first_kf_Id, related_kf_ids, pix_num: int):
        flat, kf_ids, kf_indices = self.indices_in_submap(first_kf_Id, related_kf_ids, pix_num)
        return self._gather(flat), kf_ids, kf_indices

    def indices_in_submap(self, first_kf_Id, related_kf_ids, pix_num: int, sample_range=None):
        """The host half of ``sample_rays_in_submap``: the python-``random`` draws and the index arithmetic, without
        the gather -> (flat row indices into the database, kf_ids, kf_indices), all CPU int64.  A producer thread
        calls this ahead of time; the rows are gathered later, inside the captured iteration.
        sample_range(n, k): a stand-in for ``torch.tensor(random.sample(range(n), k))`` that draws the same indices
        from the same generator (mipsfusion_amd.hostrng's session over python's global generator)."""
        if sample_range is None:
            def sample_range(n, k):
                return torch.tensor(random.sample(range(n), k))
        R = self.num_rays_to_save
        n_rel = related_kf_ids.shape[0]
        n_first = max(pix_num // n_rel, pix_num // 10)
        idx_first = sample_range(R, n_first)
        first = int(first_kf_Id)
        flat = [first * R + idx_first]
        kf_indices = [torch.zeros_like(idx_first)]
        kf_ids = [torch.ones_like(idx_first) * first_kf_Id]
        if n_rel > 1:
            tail_flat, tail_indices, tail_ids = [], [], []
            if n_rel > 2:
                last = related_kf_ids[-1]
                n_last = max(pix_num // n_rel, pix_num // 5)
                idx_last = sample_range(R, n_last)
                tail_flat, tail_indices = [int(last) * R + idx_last], [torch.ones_like(idx_last) * (n_rel - 1)]
                tail_ids = [torch.ones_like(idx_last) * last]
                other_ids, n_other, n_other_kf = related_kf_ids[1:-1], pix_num - n_first - n_last, n_rel - 2
            else:
                other_ids, n_other, n_other_kf = related_kf_ids[1:], pix_num - n_first, n_rel - 1
            idx_other = sample_range(n_other_kf * R, n_other)
            o_indices = torch.div(idx_other, R, rounding_mode="floor")
            o_ids = other_ids[o_indices]
            flat += [o_ids.to(torch.int64) * R + (idx_other - o_indices * R)] + tail_flat
            kf_indices += [o_indices + 1] + tail_indices
            kf_ids += [o_ids] + tail_ids
        return torch.cat(flat), torch.cat(kf_ids), torch.cat(kf_indices)
