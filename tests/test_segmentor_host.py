"""Host logic of REAL's segmentor contract and logit basket on CPU tensors, against the fixture captured from the
reference's own DefaultSegmentorSAM_Image + trainer statement (tests/golden/segmentor_sam.npz, generator:
tests/golden/make_golden.py::gen_segmentor; reference pointcept/models/default.py:15-76,
pointcept/engines/train_sam_real.py:229-234,286-291).  The backbone here is a stand-in that returns the fixture's
logits (the HIP backbone runs in tests/test_gpu_segmentor.py)."""
import numpy as np
import pytest
import torch

from ao_amd.ptv2.basket import LogitBasket
from ao_amd.ptv2.segmentor import DefaultSegmentor, DefaultSegmentorSAM_Image


class _FixedBackbone(torch.nn.Module):
    def __init__(self, logits):
        super().__init__()
        self.logits = torch.nn.Parameter(logits.clone())

    def forward(self, input_dict):
        return self.logits


def _fixture(golden):
    g = golden("segmentor_sam.npz")
    m = golden("ptv2_s3dis.npz")
    logits = torch.from_numpy(np.concatenate([g["seg_logits_0"], g["seg_logits_1"]]))
    batch = dict(offset=torch.from_numpy(m["offset"]), segment=torch.from_numpy(m["label"]),
                 scene_id=[str(s) for s in g["scene_id"]], instance=torch.from_numpy(g["instance"]))
    return g, logits, batch


@pytest.mark.parametrize("host_bounds", [False, True])
def test_sam_image_segmentor_contract(golden, host_bounds):
    g, logits, batch = _fixture(golden)
    seg = DefaultSegmentorSAM_Image(_FixedBackbone(logits), criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0,
                                                                           ignore_index=-1)]).train()
    if host_bounds:
        batch["offset_host"] = batch["offset"].tolist()
    out, seg_dict = seg(batch)
    assert list(out) == ["loss"] and abs(float(out["loss"]) - float(g["loss_train"])) < 1e-6
    assert list(seg_dict) == [str(k) for k in g["keys"]] == ["Area_1_office_3", "Area_4_hallway_11"]
    for i, k in enumerate(seg_dict):
        lg, ids = seg_dict[k]
        assert not lg.requires_grad
        assert torch.equal(lg, torch.from_numpy(g["seg_logits_%d" % i])) and torch.equal(ids, torch.from_numpy(g["seg_ids_%d" % i]))
    out["loss"].backward()  # the loss still reaches the backbone
    assert seg.backbone.logits.grad is not None
    seg.eval()
    ev = seg(batch)
    assert sorted(ev) == [str(k) for k in g["eval_keys"]] and abs(float(ev["loss"]) - float(g["loss_train"])) < 1e-6
    te = seg({k: v for k, v in batch.items() if k != "segment"})
    assert sorted(te) == [str(k) for k in g["test_keys"]]


def test_criteria_list_semantics(golden):
    g, logits, batch = _fixture(golden)
    half = DefaultSegmentor(_FixedBackbone(logits), criteria=[dict(type="CrossEntropyLoss", loss_weight=0.5, ignore_index=-1)]).eval()
    full = DefaultSegmentor(_FixedBackbone(logits)).eval()
    assert abs(float(half(batch)["loss"]) - 0.5 * float(full(batch)["loss"])) < 1e-7
    two = DefaultSegmentor(_FixedBackbone(logits), criteria=[dict(type="CrossEntropyLoss"), dict(type="CrossEntropyLoss", loss_weight=0.5)]).eval()
    assert abs(float(two(batch)["loss"]) - 1.5 * float(full(batch)["loss"])) < 1e-6  # Criteria sums its entries
    with pytest.raises(NotImplementedError):
        DefaultSegmentor(_FixedBackbone(logits), criteria=[dict(type="LovaszLoss")])


def test_basket_reproduces_the_trainer_statement(golden):
    g, logits, batch = _fixture(golden)
    keys = [str(k) for k in g["keys"]]
    basket = LogitBasket(dict(zip(keys, g["scene_points"].tolist())), 13, slots=2, max_rows=1024)  # forces a slot to grow
    seg_dict = {k: (torch.from_numpy(g["seg_logits_%d" % i]), torch.from_numpy(g["seg_ids_%d" % i])) for i, k in enumerate(keys)}
    basket.put(seg_dict)
    basket.flush()
    for i, k in enumerate(keys):
        assert np.array_equal(basket[k], g["basket_%d" % i])
    assert sorted(basket.as_dict()) == sorted(keys) and "nope" not in basket


def test_basket_later_steps_win_and_backpressure():
    rng = np.random.default_rng(0)
    basket = LogitBasket({"a": 500, "b": 300}, 4, slots=2, max_rows=64)
    ref = {"a": np.full((500, 4), -100.0, np.float32), "b": np.full((300, 4), -100.0, np.float32)}
    for step in range(40):  # far more steps than slots: the ring is reused, order must be step order
        d = {}
        for k, n in (("a", 500), ("b", 300)):
            ids = rng.choice(n, size=rng.integers(1, 50), replace=False)
            lg = rng.normal(size=(len(ids), 4)).astype(np.float32)
            ref[k][ids] = lg
            d[k] = (torch.from_numpy(lg), torch.from_numpy(ids))
        basket.put(d)
    basket.flush()
    assert basket.puts == 40
    for k in ref:
        assert np.array_equal(basket[k], ref[k])
    basket.close()


def test_basket_errors_and_merge():
    basket = LogitBasket({"a": 10}, 3)
    with pytest.raises(KeyError):
        basket.put({"zz": (torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64))})
    with pytest.raises(ValueError):
        basket.put({"a": (torch.zeros(2, 5), torch.zeros(2, dtype=torch.int64))})
    basket.put({"a": (torch.ones(1, 3), torch.tensor([99]))})  # id outside the scene: fails in the worker
    with pytest.raises(RuntimeError):
        basket.flush()
    # rank-0 merge of another rank's basket (train_sam_real.py:286-291)
    mine, other = LogitBasket({"a": 4}, 2), LogitBasket({"a": 4}, 2)
    mine.put({"a": (torch.full((1, 2), 1.0), torch.tensor([0]))})
    other.put({"a": (torch.full((2, 2), 2.0), torch.tensor([0, 3]))})
    mine.flush(), other.flush()
    mine.merge(other)
    assert mine["a"].tolist() == [[2.0, 2.0], [-100.0, -100.0], [-100.0, -100.0], [2.0, 2.0]]


def test_collate_keeps_scene_bounds_on_the_host():
    from ao_amd.ptv2.transform import Collect, point_collate

    parts = [dict(coord=torch.zeros(5, 3), segment=torch.zeros(5, dtype=torch.int64)),
             dict(coord=torch.zeros(7, 3), segment=torch.zeros(7, dtype=torch.int64))]
    batch = point_collate([Collect(keys=("coord", "segment"))(p) for p in parts])
    assert batch["offset"].tolist() == [5, 12] and batch["offset_host"] == [5, 12]
