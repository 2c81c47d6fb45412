"""Device-side counterparts of the three input-pipeline transforms that set the size of a training scene
(SURVEY.md §8f row 3): GridSample, SphereCrop, Collect + the point collate.

Same constructor arguments, dictionary keys and results as pointcept/datasets/transform.py:27-50 (Collect),
:769-897 (GridSample), :899-999 (SphereCrop) and pointcept/datasets/utils.py:14-54 (collate), but on CUDA tensors:
the reference runs these in numpy on 16 CPU workers per GPU; here a raw scan is voxelised and cropped on the GPU
that trains on it (voxel keys / squared distances: ao_amd/csrc/dataops.hip; sort / run-length: torch = rocPRIM).

Where the reference's result depends on numpy's unstable argsort (which point of a voxel a given draw selects, the
order of equidistant points) the order here is the stable one -- ascending original index among equals.  Random
choices come from a torch.Generator (or are passed in), not from numpy's global RNG.
"""
import torch

from .. import _lib

_SIGN = -(1 << 63)


def _dev_f32(x):
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("ao_amd.ptv2.transform works on CUDA tensors (no CPU fallback)")
    return x.contiguous().float()


class GridSample:
    def __init__(self, grid_size=0.05, hash_type="fnv", mode="train", keys=("coord", "color", "normal", "segment"),
                 return_discrete_coord=False, return_min_coord=False, return_displacement=False,
                 project_displacement=False):
        assert mode in ["train", "test"]
        self.grid_size, self.ravel, self.mode, self.keys = grid_size, hash_type != "fnv", mode, keys
        self.return_discrete_coord, self.return_min_coord = return_discrete_coord, return_min_coord
        self.return_displacement, self.project_displacement = return_displacement, project_displacement

    def _grid(self):
        g = self.grid_size
        g = [float(g)] * 3 if not isinstance(g, (list, tuple)) else [float(v) for v in g]
        assert len(g) == 3
        return g

    def voxelise(self, coord):
        """(idx_sort, start, count, cell - min, min cell): points ordered by voxel key (unsigned, as np.argsort of the
        uint64 keys), run start and length of every voxel."""
        coord = _dev_f32(coord)
        n = coord.shape[0]
        cell = torch.empty((n, 3), dtype=torch.int32, device=coord.device)
        rng = torch.empty(6, dtype=torch.int32, device=coord.device)
        key = torch.empty(n, dtype=torch.int64, device=coord.device)
        g = self._grid()
        rc = _lib.lib().grid_sample_keys_hip_launcher(n, coord.data_ptr(), g[0], g[1], g[2], int(self.ravel), cell.data_ptr(),
                                                      rng.data_ptr(), key.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "grid_sample_keys_hip_launcher")
        skey, idx_sort = torch.sort(key ^ _SIGN, stable=True)  # signed order of key ^ 2^63 == unsigned order of key
        _, count = torch.unique_consecutive(skey, return_counts=True)
        start = torch.cumsum(count, 0) - count
        lo = rng[:3].long()
        return idx_sort, start, count, cell.long() - lo, lo, (skey ^ _SIGN)

    def _extras(self, data_dict, coord, cell, lo):
        out = {}
        g = torch.tensor(self._grid(), dtype=torch.float32, device=coord.device)
        if self.return_min_coord:
            out["min_coord"] = (lo.float() * g).reshape(1, 3)
        if self.return_displacement:
            disp = coord / g - cell.float() - 0.5  # transform.py:822: against the min-shifted cell, as the reference
            if self.project_displacement:
                disp = (disp * data_dict["normal"]).sum(-1, keepdim=True)
            out["displacement"] = disp
        return out

    def __call__(self, data_dict, generator=None, draws=None):
        assert "coord" in data_dict.keys()
        coord = _dev_f32(data_dict["coord"])
        idx_sort, start, count, cell, lo, _ = self.voxelise(coord)
        extras = self._extras(data_dict, coord, cell, lo)
        if self.mode == "train":
            if draws is None:
                # transform.py:805: randint(0, count.max(), count.size) % count
                draws = torch.randint(0, int(count.max()), (count.numel(),), device=coord.device, generator=generator)
            draws = torch.as_tensor(draws, device=coord.device).long()
            idx_unique = idx_sort[start + draws % count]
            if "sampled_index" in data_dict:
                # transform.py:807-815 (ScanNet data-efficient): the labelled points are kept whatever the draw selected; the
                # selection becomes the SORTED union (np.unique) and sampled_index is re-expressed in the new numbering
                sampled = torch.as_tensor(data_dict["sampled_index"], device=coord.device).long()
                idx_unique = torch.unique(torch.cat([idx_unique, sampled]))
                mask = torch.zeros(coord.shape[0], dtype=torch.bool, device=coord.device)
                mask[sampled] = True
                data_dict["sampled_index"] = torch.nonzero(mask[idx_unique]).reshape(-1)
            if self.return_discrete_coord:
                data_dict["discrete_coord"] = cell[idx_unique]
            if self.return_min_coord:
                data_dict["min_coord"] = extras["min_coord"]
            if self.return_displacement:
                data_dict["displacement"] = extras["displacement"][idx_unique]
            for key in self.keys:
                data_dict[key] = data_dict[key][idx_unique]
            return data_dict
        parts = []
        for i in range(int(count.max())):
            idx_part = idx_sort[start + i % count]
            part = dict(index=idx_part)
            if self.return_discrete_coord:
                part["discrete_coord"] = cell[idx_part]
            if self.return_min_coord:
                part["min_coord"] = extras["min_coord"]
            if self.return_displacement:
                data_dict["displacement"] = extras["displacement"][idx_part]  # transform.py:848 writes it to data_dict
            for key in data_dict.keys():
                part[key] = data_dict[key][idx_part] if key in self.keys else data_dict[key]
            parts.append(part)
        return parts


class SphereCrop:
    CROPPED = ("coord", "origin_coord", "discrete_coord", "color", "normal", "segment", "instance", "displacement",
               "strength")  # transform.py:982-999

    def __init__(self, point_max=80000, sample_rate=None, mode="random"):
        assert mode in ["random", "center", "all"]
        self.point_max, self.sample_rate, self.mode = point_max, sample_rate, mode

    @staticmethod
    def dist2(coord, centre):
        """Squared distances of every point to `centre` (3 floats on the device), rounded as numpy's
        np.sum(np.square(coord - centre), 1) (ao_amd/csrc/dataops.hip)."""
        coord = _dev_f32(coord)
        n = coord.shape[0]
        d2 = torch.empty(n, dtype=torch.float32, device=coord.device)
        centre = centre.contiguous().float()
        rc = _lib.lib().center_dist2_hip_launcher(n, coord.data_ptr(), centre.data_ptr(), d2.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "center_dist2_hip_launcher")
        return d2

    @classmethod
    def nearest(cls, coord, centre, point_max):
        """Indices of the point_max points nearest to `centre`, ascending distance."""
        return torch.sort(cls.dist2(coord, centre), stable=True)[1][:point_max]

    PART_KEYS = ("coord", "discrete_coord", "normal", "color", "displacement", "strength")  # transform.py:933-948

    def _all(self, data_dict, point_max, generator, priority):
        """mode="all" (transform.py:914-968): overlapping crops of point_max points until every point is in one.  Each crop is
        centred on the point of lowest priority; a crop raises the priority of its members by (1 - d2 / max d2)^2."""
        coord = _dev_f32(data_dict["coord"])
        n = coord.shape[0]
        if "index" not in data_dict.keys():
            data_dict["index"] = torch.arange(n, device=coord.device)
        if n <= point_max:
            part = dict(data_dict)
            part["weight"] = torch.zeros(n, dtype=torch.float64, device=coord.device)
            part["index"] = data_dict["index"]
            return [part]
        if priority is None:  # np.random.rand(n) * 1e-3
            priority = torch.rand(n, dtype=torch.float64, device=coord.device, generator=generator) * 1e-3
        coord_p = torch.as_tensor(priority, device=coord.device).double().clone()
        covered = torch.zeros(n, dtype=torch.bool, device=coord.device)
        parts = []
        while not bool(covered.all()):
            init_idx = torch.argmin(coord_p)
            d2 = self.dist2(coord, coord[init_idx])
            idx_crop = torch.sort(d2, stable=True)[1][:point_max]
            part = {key: data_dict[key][idx_crop] for key in self.PART_KEYS if key in data_dict.keys()}
            part["weight"] = d2[idx_crop]
            part["index"] = data_dict["index"][idx_crop]
            parts.append(part)
            w = part["weight"]  # (fp32 arithmetic, as numpy's on the fp32 distances; the priorities are fp64)
            coord_p[idx_crop] += torch.square(1 - w / w.max()).double()
            covered[idx_crop] = True
        return parts

    def __call__(self, data_dict, generator=None, center_index=None, priority=None):
        assert "coord" in data_dict.keys()
        n = data_dict["coord"].shape[0]
        point_max = int(self.sample_rate * n) if self.sample_rate is not None else self.point_max
        if self.mode == "all":
            return self._all(data_dict, point_max, generator, priority)
        return self._one(data_dict, n, point_max, generator, center_index)

    def _one(self, data_dict, n, point_max, generator, center_index):
        if n <= point_max:
            return data_dict
        coord = _dev_f32(data_dict["coord"])
        if center_index is None:
            center_index = (torch.randint(0, n, (1,), device=coord.device, generator=generator)[0] if self.mode == "random"
                            else n // 2)
        idx_crop = self.nearest(coord, coord[center_index], point_max)
        for key in self.CROPPED:
            if key in data_dict.keys():
                data_dict[key] = data_dict[key][idx_crop]
        return data_dict


class Collect:
    def __init__(self, keys, offset_keys_dict=None, **kwargs):
        self.keys = [keys] if isinstance(keys, str) else keys
        self.offset_keys = dict(offset="coord") if offset_keys_dict is None else offset_keys_dict
        self.kwargs = kwargs

    def __call__(self, data_dict):
        data = {key: data_dict[key] for key in self.keys}
        for key, value in self.offset_keys.items():
            data[key] = torch.tensor([data_dict[value].shape[0]])
        for name, keys in self.kwargs.items():
            data[name.replace("_keys", "")] = torch.cat([data_dict[key].float() for key in keys], dim=1)
        return data


def point_collate(batch):
    """datasets/utils.py:14-54 for a list of Collect()-ed dicts: tensors concatenated along dim 0, every '*offset*'
    entry turned into the running end index (int32 on the coord's device, what pointops expects)."""
    out = {}
    for key in batch[0]:
        vals = [d[key] for d in batch]
        if torch.is_tensor(vals[0]):
            out[key] = torch.cat(vals)
        else:
            out[key] = list(vals)
    dev = out["coord"].device if "coord" in out else None
    host = {}
    for key in out:
        if "offset" in key and torch.is_tensor(out[key]):
            ends = torch.cumsum(out[key], dim=0).int()
            if not ends.is_cuda:  # Collect() makes the counts on the host: keep the bounds there too, so that callers
                host[key + "_host"] = ends.tolist()  # that slice per scene (DefaultSegmentorSAM_Image) need no read-back
            out[key] = ends.to(dev)
    out.update(host)
    return out
