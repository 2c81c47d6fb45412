"""REAL's per-scene logit basket without a per-step synchronisation.

Reference (pointcept/engines/train_sam_real.py): the trainer holds `self.basket = {scene_key: float array
(n_points_of_the_whole_scene, num_classes)}` (:178-181, entries never seen stay at -100, :289-291) and after every
forward does, for each scene of the batch (:229-234)

    self.basket[k][ori_idx.cpu().detach().numpy()] = seg.cpu().detach().numpy()

i.e. two blocking device-to-host copies per scene per step in the middle of the step (the backward has not been issued
yet, so the GPU idles for the copy and the host for the GPU).  At epoch end the baskets feed the SAM label refinement
(:257-583), on the host.

Here `put(seg_dict)` only ENQUEUES: the step's logits and original point ids are copied into a pinned staging slot on a
side HIP stream that waits for the forward through an event (the compute stream never waits for the copy), and a
worker thread scatters a slot into the per-scene host arrays once its event has fired.  The training thread blocks only
when every staging slot is still in flight (back-pressure), or in `flush()` (epoch end, where the reference reads the
basket).  Later writes win, in step order, exactly as the reference's sequential assignments do.

With 288 GB of HBM the whole S3DIS basket (~272 rooms x ~1 M points x 13 floats = 14 GB) would also fit on the
device; the host form is kept because the consumer (SAM prompting, voting, numpy) lives on the host.
"""
import queue
import threading

import numpy as np
import torch


class LogitBasket:
    def __init__(self, scene_sizes, num_classes, fill=-100.0, slots=4, max_rows=262144, device=None, pin_scenes=False):
        """scene_sizes: {scene_key: number of points of the un-cropped scene}.  slots x max_rows x (num_classes + 2)
        floats of pinned staging memory are allocated once (max_rows = most points in one batch; grows on demand)."""
        self.num_classes, self.fill = int(num_classes), float(fill)
        self.device = torch.device(device) if device is not None else None
        self._cuda = self.device is not None and self.device.type == "cuda"
        self._arrays = {}
        for k, n in scene_sizes.items():
            if pin_scenes and self._cuda:
                t = torch.empty((int(n), self.num_classes), dtype=torch.float32, pin_memory=True)
                t.fill_(self.fill)
                self._arrays[k] = t.numpy()
            else:
                self._arrays[k] = np.full((int(n), self.num_classes), self.fill, np.float32)
        from .. import _lib

        self._scatter = _lib.lib().basket_scatter_rows_host
        self._max_rows = int(max_rows)
        self._slots = [self._new_slot() for _ in range(int(slots))]
        self._free = queue.Queue()
        for s in self._slots:
            self._free.put(s)
        self._work = queue.Queue()
        self._error = None
        self._side = torch.cuda.Stream(self.device) if self._cuda else None
        self.puts = self.waits = 0  # steps enqueued / times put() had to wait for a free slot
        self._thread = threading.Thread(target=self._drain, name="LogitBasket", daemon=True)
        self._thread.start()

    def _new_slot(self):
        pin = self._cuda
        return dict(logits=torch.empty((self._max_rows, self.num_classes), dtype=torch.float32, pin_memory=pin),
                    ids=torch.empty(self._max_rows, dtype=torch.int64, pin_memory=pin),
                    done=torch.cuda.Event() if self._cuda else None)

    # -- training thread ------------------------------------------------------------------------------------------
    def put(self, seg_dict):
        """seg_dict: {scene_key: (logits (n, C), original ids (n,))} as DefaultSegmentorSAM_Image returns it.  Returns
        immediately; the tensors may be freed by the caller (their storage is kept alive until the copy has run)."""
        if self._error is not None:
            raise RuntimeError("LogitBasket worker failed") from self._error
        rows = sum(int(v[0].shape[0]) for v in seg_dict.values())
        if rows == 0:
            return
        for k, (lg, ids) in seg_dict.items():
            if k not in self._arrays:
                raise KeyError("scene %r is not in the basket" % (k,))
            if lg.shape[0] != ids.shape[0] or lg.shape[1] != self.num_classes:
                raise ValueError("scene %r: logits %s / ids %s" % (k, tuple(lg.shape), tuple(ids.shape)))
        try:
            slot = self._free.get_nowait()
        except queue.Empty:
            self.waits += 1
            slot = self._free.get()
        if rows > slot["logits"].shape[0]:  # a larger batch than planned for: replace this slot's staging
            self._max_rows = max(rows, self._max_rows)
            slot.update(self._new_slot())
        plan, a = [], 0
        if self._cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                for k, (lg, ids) in seg_dict.items():
                    b = a + lg.shape[0]
                    slot["logits"][a:b].copy_(lg.detach(), non_blocking=True)
                    slot["ids"][a:b].copy_(ids, non_blocking=True)
                    lg.record_stream(self._side)
                    ids.record_stream(self._side)
                    plan.append((k, a, b))
                    a = b
                slot["done"].record(self._side)
        else:
            for k, (lg, ids) in seg_dict.items():
                b = a + lg.shape[0]
                slot["logits"][a:b].copy_(lg.detach())
                slot["ids"][a:b].copy_(ids)
                plan.append((k, a, b))
                a = b
        self.puts += 1
        self._work.put((slot, plan))

    def flush(self):
        """Block until everything enqueued so far is in the host arrays (epoch end, before the basket is read)."""
        self._work.join()
        if self._error is not None:
            raise RuntimeError("LogitBasket worker failed") from self._error

    def close(self):
        self.flush()
        self._work.put(None)
        self._thread.join()

    # -- worker thread --------------------------------------------------------------------------------------------
    def _drain(self):
        while True:
            item = self._work.get()
            if item is None:
                self._work.task_done()
                return
            slot, plan = item
            try:
                if slot["done"] is not None:
                    slot["done"].synchronize()  # blocks this thread only
                # the reference's statement `basket[k][ori_idx] = seg` (train_sam_real.py:234) as one native call per
                # scene (ao_amd/csrc/abi.hip: basket_scatter_rows_host).  ctypes drops the interpreter lock for the
                # call, so this thread does not stall the training thread's kernel launches: numpy's fancy assignment
                # holds the lock (measured +0.7 ms on a 15 ms step), torch's index_copy_ from a second thread starts a
                # second OpenMP team (measured 53 ms per step on a 128-core host).
                lg_ptr, id_ptr = slot["logits"].data_ptr(), slot["ids"].data_ptr()
                for k, a, b in plan:
                    arr = self._arrays[k]
                    rc = self._scatter(arr.ctypes.data, arr.shape[0], id_ptr + 8 * a, lg_ptr + 4 * self.num_classes * a,
                                       b - a, self.num_classes)
                    if rc != 0:
                        raise IndexError("scene %r: an original point id is outside [0, %d)" % (k, arr.shape[0]))
            except BaseException as e:  # surfaced by the next put() / flush()
                self._error = e
            finally:
                self._free.put(slot)
                self._work.task_done()

    # -- the reference's dict view (read after flush()) -------------------------------------------------------------
    def __getitem__(self, k):
        return self._arrays[k]

    def __contains__(self, k):
        return k in self._arrays

    def keys(self):
        return self._arrays.keys()

    def items(self):
        return self._arrays.items()

    def as_dict(self):
        """{scene_key: ndarray} -- what the reference pickles / merges across ranks (:266-291)."""
        return dict(self._arrays)

    def merge(self, other):
        """Rank-0 merge of another rank's basket (:286-291): entries the other rank has seen overwrite ours."""
        for k, v in (other.items() if hasattr(other, "items") else other):
            seen = v != self.fill
            self._arrays[k][seen] = v[seen]
