"""oracle/host_ref.py -- CPU (numpy) restatement of the host-side integer work either side of the training step
(SURVEY.md §8f rows 1, 3, 4).  TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/golden/make_golden_host.py,
never by ao_amd/.

Pinned against the reference's own Python (imported in the build container by tests/golden/make_golden_host.py,
outputs committed as tests/golden/host_*.npz):
  * GridSample            pointcept/datasets/transform.py:769-897
  * SphereCrop            pointcept/datasets/transform.py:899-999
  * intersection_and_union pointcept/utils/misc.py:38-70
  * LR schedules          pointcept/utils/scheduler.py:14-140 (thin subclasses of torch.optim.lr_scheduler)

Two places where the reference's result is not a function of its inputs alone, and what is pinned instead:
  * `np.argsort(key)` (transform.py:799, :978) is an unstable sort: the order of points inside one voxel / of equidistant
    points is whatever numpy's introsort build produces.  The restatement uses a STABLE sort (ascending original index
    among equals); fixtures are compared on what the reference does determine (sorted unique keys, counts, the voxel of
    every selected point, the crop as a set and its distance order).
  * `coord / np.array(grid_size)` (transform.py:794) is an fp32 division under the numpy 1.x the reference ran on
    (value-based casting of the 0-d array) and an fp64 one under numpy 2.  The restatement is the fp32 form; the golden
    script passes grid_size as np.float32 so that the reference takes the same path under this container's numpy 2.2.
"""
import math

import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def fnv_hash_vec(cells):
    """transform.py:883-897: FNV64-1A over the columns, multiply first, then xor (mod 2^64)."""
    a = np.asarray(cells).astype(np.uint64)
    h = np.full(a.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(a.shape[1]):
            h = h * FNV_PRIME
            h = h ^ a[:, j]
    return h


def ravel_hash_vec(cells):
    """transform.py:865-881: ((x * (max_y + 1)) + y) * (max_z + 1) + z on min-shifted cells."""
    a = np.asarray(cells).astype(np.int64)
    a = (a - a.min(0)).astype(np.uint64)
    ext = a.max(0).astype(np.uint64) + np.uint64(1)
    keys = np.zeros(a.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(a.shape[1] - 1):
            keys = keys + a[:, j]
            keys = keys * ext[j + 1]
        keys = keys + a[:, -1]
    return keys


def grid_cells(coord, grid_size):
    """transform.py:794-797 in fp32: (cell - min, min)."""
    g = np.broadcast_to(np.asarray(grid_size, dtype=np.float32), (3,))
    scaled = np.asarray(coord, dtype=np.float32) / g
    cell = np.floor(scaled).astype(np.int64)
    lo = cell.min(0)
    return cell - lo, lo


def grid_sample_sorted(coord, grid_size, hash_type="fnv"):
    """transform.py:794-801: (idx_sort [stable], unique keys ascending as uint64, count per key, cell, min cell)."""
    cell, lo = grid_cells(coord, grid_size)
    key = fnv_hash_vec(cell) if hash_type == "fnv" else ravel_hash_vec(cell)
    idx_sort = np.argsort(key, kind="stable")
    uniq, count = np.unique(key[idx_sort], return_counts=True)
    return idx_sort, uniq, count, cell, lo


def grid_sample_train(coord, grid_size, draws=None, hash_type="fnv"):
    """transform.py:802-807: one point per voxel, `draws` = the reference's np.random.randint(0, count.max(), count.size)
    (one integer per voxel in ascending key order); None -> zeros."""
    idx_sort, uniq, count, cell, lo = grid_sample_sorted(coord, grid_size, hash_type)
    draws = np.zeros(count.size, dtype=np.int64) if draws is None else np.asarray(draws, dtype=np.int64)
    start = np.cumsum(np.insert(count, 0, 0)[:-1])
    return idx_sort[start + draws % count]


def grid_sample_test(coord, grid_size, hash_type="fnv"):
    """transform.py:830-837: count.max() parts, part i takes element (i mod count) of every voxel."""
    idx_sort, uniq, count, cell, lo = grid_sample_sorted(coord, grid_size, hash_type)
    start = np.cumsum(np.insert(count, 0, 0)[:-1])
    return [idx_sort[start + i % count] for i in range(int(count.max()))]


def sphere_crop(coord, point_max, center_index):
    """transform.py:966-981: the point_max points nearest to coord[center_index], ascending squared distance."""
    c = np.asarray(coord, dtype=np.float32)
    if c.shape[0] <= point_max:
        return np.arange(c.shape[0])
    d2 = np.sum(np.square(c - c[center_index]), 1)
    return np.argsort(d2, kind="stable")[:point_max]


def center_dist2(coord, center):
    c = np.asarray(coord, dtype=np.float32)
    return np.sum(np.square(c - np.asarray(center, dtype=np.float32)), 1)


def intersection_and_union(output, target, k, ignore_index=-1):
    """misc.py:38-70: (intersection, union, target) class histograms; output is masked where target is ignored."""
    o = np.asarray(output).reshape(-1).astype(np.int64).copy()
    t = np.asarray(target).reshape(-1).astype(np.int64)
    o[t == ignore_index] = ignore_index
    inter = np.bincount(o[(o == t) & (o >= 0) & (o < k)], minlength=k)[:k]
    area_o = np.bincount(o[(o >= 0) & (o < k)], minlength=k)[:k]
    area_t = np.bincount(t[(t >= 0) & (t < k)], minlength=k)[:k]
    return inter, area_o + area_t - inter, area_t


def miou(intersection, union, target):
    """evaluator.py:165-171."""
    iou = intersection / (union + 1e-10)
    acc = intersection / (target + 1e-10)
    return float(np.mean(iou)), float(np.mean(acc)), float(np.sum(intersection) / (np.sum(target) + 1e-10))


# ---------------------------------------------------------------- LR schedules --
def lr_curve(kind, base_lr, total_steps, steps, **kw):
    """Learning rate in effect for optimizer step 0, 1, ..., steps-1 (the scheduler is stepped once after every
    optimizer step, engines/train.py:184-196); OneCycleLR additionally returns the beta1 / momentum curve."""
    lrs, moms = [], []
    for s in range(steps):
        if kind == "MultiStepLR":  # scheduler.py:14-31; torch compares the integer step with the FLOAT milestones exactly
            miles = [r * total_steps for r in kw["milestones"]]
            lrs.append(base_lr * kw.get("gamma", 0.1) ** sum(1 for t in range(1, s + 1) if float(t) in miles))
        elif kind == "MultiStepWithWarmupLR":  # scheduler.py:34-68
            miles = [r * total_steps for r in kw["milestones"]]
            gamma, wr, ws = kw.get("gamma", 0.1), kw.get("warmup_rate", 0.05), kw.get("warmup_scale", 1e-6)
            f = 1.0
            for t in miles:
                if s < t:
                    break
                f *= gamma
            w = 1 - (1 - s / wr / total_steps) * (1 - ws) if s <= wr * total_steps else 1.0
            lrs.append(base_lr * w * f)
        elif kind == "PolyLR":  # scheduler.py:71-79
            lrs.append(base_lr * (1 - s / (total_steps + 1)) ** kw.get("power", 0.9))
        elif kind == "ExpLR":  # scheduler.py:82-90
            lrs.append(base_lr * kw.get("gamma", 0.9) ** (s / total_steps))
        elif kind == "CosineAnnealingLR":  # scheduler.py:93-103 (closed form of torch's recurrence)
            eta = kw.get("eta_min", 0.0)
            lrs.append(eta + (base_lr - eta) * (1 + math.cos(math.pi * s / total_steps)) / 2)
        elif kind == "OneCycleLR":  # scheduler.py:106-140, two-phase, cos or linear anneal
            max_lr = kw["max_lr"]
            init = max_lr / kw.get("div_factor", 25.0)
            low = init / kw.get("final_div_factor", 1e4)
            m_hi, m_lo = kw.get("max_momentum", 0.95), kw.get("base_momentum", 0.85)
            e1 = float(kw.get("pct_start", 0.3) * total_steps) - 1
            e2 = total_steps - 1

            def anneal(a, b, p):
                if kw.get("anneal_strategy", "cos") == "cos":
                    return b + (a - b) / 2.0 * (math.cos(math.pi * p) + 1)
                return (b - a) * p + a

            if s <= e1:
                p = s / e1
                lrs.append(anneal(init, max_lr, p)); moms.append(anneal(m_hi, m_lo, p))
            else:
                p = (s - e1) / (e2 - e1)
                lrs.append(anneal(max_lr, low, p)); moms.append(anneal(m_lo, m_hi, p))
        else:
            raise KeyError(kind)
    return (np.asarray(lrs), np.asarray(moms)) if kind == "OneCycleLR" else np.asarray(lrs)
