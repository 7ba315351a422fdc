"""Queued, batched serving in front of ONE shared model -- the request front-end of SURVEY 8(f)-3.

The reference serves requests by letting Gradio's worker threads call `action_inference` on a global
model with no lock and one request = one `sampler.sample` of `n_sample_image` samples (app.py:229-277,
405-410).  Here every request goes through a FIFO; one worker thread owns the device and

  * coalesces compatible queued requests (same output size, step count, guidance scale, eta, control /
    unconditional-context kind) into ONE DDIM batch: samples are independent on this path, every sample
    carries its own SeeCoder context in the cross-attention K / V^T cache, so four single-image requests
    cost one batch-4 loop instead of four batch-1 loops (the UNet at batch 2 fills 1/4 of the chip);
  * applies weight hot swaps (`load`, the per-request tag switches of app.py:217-222) strictly in queue
    order between batches -- the packed-weight caches and captured hipGraphs are keyed on the parameter
    versions, so the next batch re-packs / re-captures;
  * returns packed uint8 HWC images (`ToPILImage`'s arithmetic on the device, pfd_image_u8_f16) or the
    float images, through `concurrent.futures.Future`s.
"""
import queue
import threading
from concurrent.futures import Future

import torch

from .pipeline import PromptFreePipeline, shard_xT


class _Request:
    __slots__ = ("image", "n", "height", "width", "steps", "scale", "eta", "seed", "control", "uncond", "as_uint8",
                 "future")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def key(self):
        """requests with equal keys can share one DDIM batch"""
        return (self.height, self.width, self.steps, float(self.scale), float(self.eta), self.control is None,
                self.uncond is None, bool(self.as_uint8))

    def shareable(self):
        """eta > 0 draws its noise from the global RNG inside the loop (ddim.py:166): batched with other requests the
        draw -- and so the result for a given seed -- would depend on the company; such requests run alone, seeded"""
        return float(self.eta) == 0.0 and self.control is None


class PromptFreeServer:
    def __init__(self, net, use_graph=True, max_batch=8, max_wait_s=0.0):
        self.net = net
        self.pipe = PromptFreePipeline(net)
        self.pipe.enable_graph(use_graph)
        self.max_batch, self.max_wait_s = int(max_batch), float(max_wait_s)
        self._q = queue.Queue()
        self._stop = False
        self.batches = []                  # sizes of the DDIM batches run so far (observability / tests)
        self._worker = threading.Thread(target=self._run, name="pfd-server", daemon=True)
        self._worker.start()

    # ---- client side (any thread) -------------------------------------------------------------------
    def submit(self, image, n_samples=1, height=512, width=512, steps=50, scale=2.0, eta=0.0, seed=20,
               control=None, uncond=None, as_uint8=True):
        """image [1,3,h,w] in [0,1] (any size); control [1,3,H,W] or None; uncond [1,148,768] or None (zeros).
        -> Future of uint8 [n,H,W,3] (as_uint8) or float [n,3,H,W] images"""
        if self._stop:
            raise RuntimeError("server is closed")
        if height % 64 or width % 64:
            raise ValueError("output size must be a multiple of 64 (app.py:226-227)")
        # everything a coalesced batch would first touch inside _generate is checked HERE, on the caller's thread: a
        # malformed tensor must fail its own request, not the batch it was merged into
        if n_samples < 1 or n_samples > self.max_batch:
            raise ValueError(f"n_samples must be in [1, {self.max_batch}]")
        if steps < 1:
            raise ValueError("steps must be >= 1")
        if not (torch.is_tensor(image) and image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 3 and
                image.is_floating_point() and min(image.shape[2:]) >= 32):
            raise ValueError("image must be a float tensor [1, 3, h, w] in [0, 1]")
        if uncond is not None and not (torch.is_tensor(uncond) and uncond.is_floating_point() and
                                       tuple(uncond.shape) == (1, 148, 768)):
            raise ValueError("uncond must be a float tensor [1, 148, 768]")
        if control is not None and not (torch.is_tensor(control) and control.is_floating_point() and
                                        tuple(control.shape) == (1, 3, int(height), int(width))):
            raise ValueError(f"control must be a float tensor [1, 3, {height}, {width}]")
        r = _Request(image=image, n=int(n_samples), height=int(height), width=int(width), steps=int(steps),
                     scale=scale, eta=eta, seed=int(seed), control=control, uncond=uncond, as_uint8=as_uint8,
                     future=Future())
        self._q.put(r)
        return r.future

    def load(self, kind, path):
        """queue a weight hot swap ('ctx' | 'diffuser' | 'ctl', app.py:139-161); applied in order"""
        f = Future()
        self._q.put(("load", kind, path, f))
        return f

    def call(self, fn):
        """queue an arbitrary action on the model (e.g. attaching a PPE_MLP, app.py:166-177); applied in order"""
        f = Future()
        self._q.put(("call", fn, None, f))
        return f

    def close(self):
        self._stop = True
        self._q.put(None)
        self._worker.join()
        while True:      # a submit() that raced past the _stop check sits behind the sentinel: fail it, never leave it pending
            try:
                item = self._q.get_nowait()
            except queue.Empty:
                break
            f = item.future if isinstance(item, _Request) else (item[3] if isinstance(item, tuple) else None)
            if f is not None and not f.done():
                f.set_exception(RuntimeError("server is closed"))

    # ---- worker -------------------------------------------------------------------------------------
    def _run(self):
        pending = None
        while True:
            item = pending if pending is not None else self._q.get()
            pending = None
            if item is None:
                break
            if isinstance(item, tuple):
                self._control(item)
                continue
            batch, total = [item], item.n
            while total < self.max_batch and item.shareable():   # coalesce what is already queued (or arrives within max_wait_s)
                try:
                    nxt = self._q.get(timeout=self.max_wait_s) if self.max_wait_s > 0 else self._q.get_nowait()
                except queue.Empty:
                    break
                if nxt is None or isinstance(nxt, tuple) or nxt.key() != item.key() or \
                        total + nxt.n > self.max_batch or nxt.control is not None:
                    pending = nxt                   # order is preserved: it starts the next round
                    if nxt is None:
                        pending = None
                        self._q.put(None)
                    break
                batch.append(nxt)
                total += nxt.n
            try:
                outs = self._generate(batch)
                for r, o in zip(batch, outs):
                    r.future.set_result(o)
            except BaseException as e:   # noqa: BLE001 -- delivered to the callers, the worker keeps serving
                if len(batch) == 1:
                    if not batch[0].future.done():
                        batch[0].future.set_exception(e)
                    continue
                # a coalesced batch failed: run its requests one by one so that only the one that causes the error
                # receives it (the others would otherwise fail with somebody else's exception)
                for r in batch:
                    if r.future.done():
                        continue
                    try:
                        r.future.set_result(self._generate([r])[0])
                    except BaseException as e1:   # noqa: BLE001
                        r.future.set_exception(e1)

    def _control(self, item):
        op, a, b, f = item
        try:
            if op == "load":
                from . import weights_io
                fn = {"ctx": weights_io.load_ctx, "diffuser": weights_io.load_diffuser, "ctl": weights_io.load_ctl}[a]
                f.set_result(fn(self.net, b))
            else:
                f.set_result(a(self.net))
        except BaseException as e:   # noqa: BLE001
            f.set_exception(e)

    @torch.no_grad()
    def _generate(self, batch):
        """one DDIM batch for a list of compatible requests; returns one image tensor per request"""
        from .hip.ops import DEVICE_LOCK
        r0 = batch[0]
        dev = self.net.device
        with DEVICE_LOCK:
            if float(r0.eta) != 0.0:       # runs alone (shareable()): its noise stream is a function of ITS seed only
                torch.manual_seed(r0.seed)
            conds, xts, unconds = [], [], []
            for r in batch:
                c, z = self.pipe.encode_reference(r.image.to(dev), r.n)
                conds.append(c)
                unconds.append(z if r.uncond is None else r.uncond.to(dev).to(c.dtype).expand(r.n, -1, -1))
                xts.append(shard_xT(r.n, r.height, r.width, r.seed, 0, 1))      # each request keeps ITS x_T stream
            cond, uncond, xT = torch.cat(conds), torch.cat(unconds), torch.cat(xts).to(dev)
            c_info = {'type': 'image', 'conditioning': cond, 'unconditional_conditioning': uncond,
                      'unconditional_guidance_scale': r0.scale}
            if r0.control is not None:
                c_info['control'] = r0.control.to(dev)
            x, _ = self.pipe.sampler.sample(steps=r0.steps, shape=list(xT.shape), x_info={'type': 'image', 'xt': xT},
                                            c_info=c_info, eta=r0.eta, verbose=False)
            img = self.net.vae_decode(x, 'image', out_uint8=True) if r0.as_uint8 else self.net.vae_decode(x, 'image')
            self.batches.append(int(xT.shape[0]))
        outs, off = [], 0
        for r in batch:
            outs.append(img[off:off + r.n])
            off += r.n
        return outs
