"""Host -> HBM hand-off of the batch producer: the NEXT batch is pinned and copied on a dedicated HIP stream while the
current step computes, so the PCIe transfer leaves the critical path of the step.

The reference moves every batch synchronously at the top of the step (`batch = tuple(t.to(device) ...)`,
run_multimodal_fcmf.py:440-446, run_pretraining_fcmf.py:297-303) from pageable DataLoader memory, then converts the float64
ROI crops on the device (`roi_img_features.float()`, :447).  A B=64 batch of precomputed features is 312 MB in float32
(156 MB from the bf16 `FeatureCache`): 6-24 ms of a 42 ms step when it is not overlapped.

  * a worker thread pulls batches from the loader and STAGES them in page-locked memory: a ring of reusable pinned buffers
    per batch field (allocated once -- `pin_memory()` per batch page-locks 300 MB every step, measured 60 ms against a 38 ms
    step), filled by a few copy threads (one thread's memcpy moves ~10 GB/s: 30 ms for a float32 B=64 batch); none of it
    sits in the thread that launches kernels;
  * the main thread issues the `non_blocking` copies of batch i+1 on `copy_stream` right after handing out batch i, and
    makes the compute stream wait for batch i's copy event only when batch i is about to be used;
  * dtype hand-off on the device, on the copy stream: float64 pixel crops -> float32 (what the reference's `.float()` does;
    the ROI BOXES stay float64 -- the parity mode does the box geometry in float64, roi_modeling.py:79-138), bf16 features
    stay bf16 (the MFMA path consumes them as they are);
  * tensors are `record_stream`ed on the compute stream: the caching allocator will not recycle them while the step runs.
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import torch

_COPY_THREADS = 8
_PARALLEL_MIN_BYTES = 8 << 20


class DevicePrefetcher:
    def __init__(self, loader, device, depth=2, float64_pixels_to_float32=True, float32_fields=None):
        """loader: any iterable of tuples / dicts of CPU tensors (DataLoader, SyntheticBatches, a generator);
        depth: how many batches may wait pinned on the host;
        float32_fields: tuple positions / dict keys whose float64 tensors become float32 on the device (the drivers name the
        ROI crops: element 1 of the reference's batch tuple).  None: decided by shape -- pixel crops are [..., 3, H, W] tensors
        of at least 5 dimensions with more than 4 columns; the box tensor [B, num_imgs, num_rois, 4] never qualifies, whatever
        num_imgs is (round-3 advisor finding: with --num_imgs 3 the old rule `shape[-3] == 3` cast the boxes too)"""
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.f64_to_f32 = float64_pixels_to_float32
        self.f32_fields = None if float32_fields is None else set(float32_fields)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bytes_copied = 0
        # pinned staging ring: (field, shape, dtype) -> [buffers]; a buffer is reused only after the upload that read it has
        # finished (its event, recorded by the consumer thread in _upload)
        self._ring, self._ring_pos, self._busy, self._mine = {}, {}, {}, set()
        self._ring_len = self.depth + 4
        self._pool = None

    def __len__(self):
        return len(self.loader)

    # ---- host side (worker thread) ---------------------------------------------------------------------------
    def _pin(self, x, field=None):
        """pageable CPU tensor -> a pinned buffer of the staging ring holding the same values"""
        if not (torch.is_tensor(x) and not x.is_cuda and not x.is_pinned() and x.numel() > 0):
            return x
        x = x.contiguous()
        key = (field, tuple(x.shape), x.dtype)
        ring = self._ring.setdefault(key, [])
        pos = self._ring_pos.get(key, 0)
        if len(ring) < self._ring_len:
            ring.append(torch.empty(x.shape, dtype=x.dtype).pin_memory())
            self._mine.add(ring[-1].data_ptr())
        buf = ring[pos % len(ring)]
        self._ring_pos[key] = pos + 1
        ev = self._busy.pop(buf.data_ptr(), None)
        if ev is not None:
            ev.synchronize()                      # (the upload that last read this buffer; long finished in steady state)
        nbytes = x.numel() * x.element_size()
        if nbytes < _PARALLEL_MIN_BYTES:
            buf.copy_(x)
        else:                                     # torch's copy_ drops the GIL: a few threads fill disjoint row ranges
            if self._pool is None:
                self._pool = ThreadPoolExecutor(_COPY_THREADS, thread_name_prefix="prefetch-copy")
            src, dst, n = x.view(-1), buf.view(-1), x.numel()
            step = (n + _COPY_THREADS - 1) // _COPY_THREADS
            list(self._pool.map(lambda a: dst[a:a + step].copy_(src[a:a + step]), range(0, n, step)))
        return buf

    def _map(self, batch, fn, keyed=False):
        call = (lambda k, v: fn(v, k)) if keyed else (lambda k, v: fn(v))
        if isinstance(batch, dict):
            return {k: call(k, v) for k, v in batch.items()}
        if isinstance(batch, (tuple, list)):
            return tuple(call(i, v) for i, v in enumerate(batch))
        return call(None, batch)

    def _worker(self, q, stop):
        try:
            for batch in self.loader:
                if stop.is_set():
                    return
                q.put(self._map(batch, self._pin, keyed=True))
            q.put(StopIteration)
        except BaseException as e:            # surfaces in the consumer, never swallowed
            q.put(e)

    # ---- device side (caller's thread) -----------------------------------------------------------------------
    def _is_pixels(self, y, field):
        if self.f32_fields is not None:
            return field in self.f32_fields
        return y.dim() >= 5 and y.shape[-3] == 3 and y.shape[-1] > 4

    def _to_device(self, x, field=None):
        if not torch.is_tensor(x):
            return x
        y = x.to(self.device, non_blocking=True)
        self.bytes_copied += x.numel() * x.element_size()
        # pixel crops [.., 3, H, W] arrive in float64 from the reference's dataset (vimacsa_dataset.py:175-199): float32 on
        # the device, as the reference's `.float()`; the box tensor [B, num_imgs, num_rois, 4] keeps its dtype
        if self.f64_to_f32 and y.dtype == torch.float64 and self._is_pixels(y, field):
            y = y.float()
        return y

    def _upload(self, host):
        with torch.cuda.stream(self.copy_stream):
            dev = self._map(host, self._to_device, keyed=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        for v in (host.values() if isinstance(host, dict) else host if isinstance(host, (tuple, list)) else (host,)):
            if torch.is_tensor(v) and v.data_ptr() in self._mine:
                self._busy[v.data_ptr()] = ev     # the staging ring may overwrite it once this copy is done
        return dev, ev, host                  # (host: the pinned source must outlive the asynchronous copy)

    def __iter__(self):
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        t = threading.Thread(target=self._worker, args=(q, stop), daemon=True)
        t.start()

        def take():
            item = q.get()
            if item is StopIteration:
                return None
            if isinstance(item, BaseException):
                raise item
            return self._upload(item)
        keep = []                             # pinned sources of copies that may still be in flight
        try:
            nxt = take()
            while nxt is not None:
                dev, ev, host = nxt
                nxt = take()                  # batch i+1 starts travelling before batch i is consumed
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for v in (dev.values() if isinstance(dev, dict) else dev if isinstance(dev, tuple) else (dev,)):
                    if torch.is_tensor(v):
                        v.record_stream(cur)
                keep.append((ev, host))
                keep = [(e, h) for e, h in keep if not e.query()]
                yield dev
        finally:
            stop.set()
            while t.is_alive():               # unblock a worker stuck on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)
