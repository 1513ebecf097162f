"""Image loader feeding the batched entry points.

The reference decodes every frame with `cv2.imread(path, cv2.IMREAD_GRAYSCALE)` inside
`get_matches` (/root/reference/libs/vo_utils.py:94-95), twice per frame, on the thread that
also runs the whole pipeline; the path list comes from `load_image_paths` (:65-71).  At the
GPU path's rate the PNG inflate (about 280 KB per KITTI frame) is the bottleneck, so decoding
is taken off the critical path:

  * a pool of decoder threads (Pillow's decoders release the GIL) fills chunks of frames,
  * each chunk lands in a page-locked (pinned) staging buffer of a small ring,
  * and is copied to the GPU on a side stream, so the copy of chunk k+1 and the decode of
    chunk k+2 overlap `vo_sequence_run_device` / `vo_pairs_run_device` on chunk k.

Decoded pixels are exactly Pillow's 8-bit gray ("L") conversion = cv2's IMREAD_GRAYSCALE for
8-bit gray PNGs; colour inputs go through Pillow's ITU-R 601 luma, which can
differ from cv2's fixed-point BGR2GRAY by one grey level -- KITTI odometry frames are gray.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def load_image_paths(filepath):
    """libs/vo_utils.py:65-71: sorted .png / .jpg files of a directory."""
    return [os.path.join(filepath, f) for f in sorted(os.listdir(filepath)) if f.endswith('.png') or f.endswith('.jpg')]


def decode_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) -> uint8 (h, w)."""
    from PIL import Image

    with Image.open(path) as im:
        if im.mode != "L":
            im = im.convert("L")
        return np.asarray(im)


def _decode_into_shm(job):
    """Worker-process side of FrameLoader(processes=N): decode one file straight into the shared staging ring."""
    from multiprocessing import shared_memory
    path, shm_name, offset, h, w = job
    im = decode_gray(path)
    if im.shape != (h, w):
        return path, im.shape
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        np.ndarray((h, w), np.uint8, buffer=shm.buf, offset=offset)[...] = im
    finally:
        shm.close()
    return None


def pack_sequence(paths, out_path):
    """Decode a sequence once into ONE uint8 [F,h,w] .npy file; FrameLoader(np.load(out_path, mmap_mode="r"), ...) then
    feeds the GPU without any decoding (PNG decode caps a 16-core host at ~5.8 k frames/s, the GPU path takes ~85 k)."""
    paths = list(paths)
    first = decode_gray(paths[0])
    out = np.lib.format.open_memmap(out_path, mode="w+", dtype=np.uint8, shape=(len(paths),) + first.shape)
    out[0] = first
    for i, p in enumerate(paths[1:], 1):
        im = decode_gray(p)
        if im.shape != first.shape:
            raise ValueError("%s: frame size %s differs from %s" % (p, im.shape, first.shape))
        out[i] = im
    out.flush()
    return out_path


class FrameLoader:
    """Iterate over a list of image files in chunks of `chunk` frames, `overlap` frames shared
    between consecutive chunks (1 for sequences: pair i needs frames i and i+1).

    device=None yields numpy arrays [n,h,w]; device="cuda:0" (or an int) yields torch uint8
    tensors on that GPU, uploaded from pinned memory on a side stream `depth` chunks ahead."""

    def __init__(self, paths, chunk=64, overlap=0, workers=None, depth=2, device=None, processes=0):
        """processes=N: decode in N worker PROCESSES (spawned, so it is safe after the GPU has been initialised) that write
        into a shared-memory ring, instead of `workers` threads.  Pillow re-takes the GIL about 30 times per KITTI PNG, so
        decoder threads stop scaling at 3-4 (measured: 1.1 k frames/s with 4 or 16 threads); processes scale with the cores."""
        self.processes = int(processes)
        if chunk <= overlap:
            raise ValueError("chunk must exceed overlap")
        # already decoded frames: a uint8 array [F,h,w] -- typically the np.load(..., mmap_mode="r") of a file written by
        # pack_sequence(); nothing is decoded, chunks go from the (page-cached) file to the pinned ring at memcpy speed
        self.frames = paths if isinstance(paths, np.ndarray) else None
        if self.frames is not None:
            if self.frames.ndim != 3 or self.frames.dtype != np.uint8:
                raise ValueError("packed frames must be a uint8 array [F,h,w]")
            paths = range(len(self.frames))
            self.processes = 0
        self.paths = list(paths)
        self.chunk, self.overlap, self.depth = int(chunk), int(overlap), max(1, int(depth))
        self.workers = workers or max(1, min(16, (os.cpu_count() or 2) - 1))
        self.device = device
        starts, s, n = [], 0, len(self.paths)
        while s < n and (not starts or s + self.overlap < n):
            starts.append(s)
            s += self.chunk - self.overlap
        self.chunks = [(a, min(a + self.chunk, n)) for a in starts]

    def __len__(self):
        return len(self.chunks)

    def __iter__(self):
        if not self.paths:
            return
        shm = None
        if self.processes > 0:
            import multiprocessing as mp
            from multiprocessing import shared_memory
            pool = mp.get_context("spawn").Pool(self.processes)
        else:
            pool = ThreadPoolExecutor(self.workers)
        try:
            h, w = self.frames.shape[1:] if self.frames is not None else decode_gray(self.paths[0]).shape
            nslots = self.depth + 2
            if self.processes > 0:
                shm = shared_memory.SharedMemory(create=True, size=nslots * self.chunk * h * w)
            torch = stream = None
            ring = []
            if self.device is not None:
                import torch
                dev = torch.device("cuda", self.device) if isinstance(self.device, int) else torch.device(self.device)
                stream = torch.cuda.Stream(device=dev)
                ring = [torch.empty((self.chunk, h, w), dtype=torch.uint8).pin_memory() for _ in range(self.depth + 1)]
                ring_ev = [None] * len(ring)

            def submit(ci):
                a, b = self.chunks[ci]
                if self.frames is not None:
                    return None
                if shm is not None:
                    base = (ci % nslots) * self.chunk * h * w
                    return pool.map_async(_decode_into_shm, [(p, shm.name, base + k * h * w, h, w) for k, p in enumerate(self.paths[a:b])])
                return [pool.submit(decode_gray, p) for p in self.paths[a:b]]

            def finish(ci, futs):
                if self.frames is not None:
                    a, b = self.chunks[ci]
                    imgs = self.frames[a:b]
                    if torch is None:
                        return np.array(imgs)
                elif shm is not None:
                    a, b = self.chunks[ci]
                    for bad in futs.get():
                        if bad is not None:
                            raise ValueError("%s: frame size %s differs from %s" % (bad[0], bad[1], (h, w)))
                    view = np.ndarray((b - a, h, w), np.uint8, buffer=shm.buf, offset=(ci % nslots) * self.chunk * h * w)
                    imgs = view if torch is not None else view.copy()    # the ring slot is reused nslots chunks later
                else:
                    imgs = [f.result() for f in futs]
                    for k, im in enumerate(imgs):
                        if im.shape != (h, w):
                            raise ValueError("%s: frame size %s differs from %s" % (self.paths[self.chunks[ci][0] + k], im.shape, (h, w)))
                if torch is None:
                    return imgs if shm is not None else np.stack(imgs)
                if isinstance(imgs, np.ndarray) and imgs.ndim == 3:
                    imgs = [imgs]                         # one block copy into the pinned buffer instead of one per frame
                slot = ci % len(ring)
                if ring_ev[slot] is not None:
                    ring_ev[slot].synchronize()      # the upload that last used this staging buffer has finished
                nfr = sum(len(x) if x.ndim == 3 else 1 for x in imgs)
                host = ring[slot][:nfr]
                k = 0
                for im in imgs:
                    if im.ndim == 3:
                        # block copy file cache -> pinned memory, split over the pool's threads (numpy copies run outside the GIL;
                        # one thread moves ~10 GB/s = 21 k KITTI frames/s, less than the GPU path consumes)
                        dst = host[k:k + len(im)].numpy()
                        step = max(1, (len(im) + self.workers - 1) // self.workers)

                        def _copy(a0, dst=dst, im=im, step=step):
                            dst[a0:a0 + step] = im[a0:a0 + step]
                        if self.processes > 0:                 # the decoder processes' pool cannot run closures: one plain copy
                            dst[...] = im
                        else:
                            list(pool.map(_copy, range(0, len(im), step)))
                        k += len(im)
                    else:
                        host[k].numpy()[...] = im
                        k += 1
                with torch.cuda.stream(stream):
                    d = host.to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                ring_ev[slot] = ev
                return d, ev

            pending = {}
            uploaded = {}
            nxt = 0
            for ci in range(len(self.chunks)):
                while nxt < len(self.chunks) and nxt <= ci + self.depth:
                    pending[nxt] = submit(nxt)
                    nxt += 1
                # decode-complete + upload issue runs one chunk ahead, so the copy of chunk ci+1 overlaps the
                # consumer's work on chunk ci
                for cj in (ci, ci + 1):
                    if cj < len(self.chunks) and cj not in uploaded:
                        uploaded[cj] = finish(cj, pending.pop(cj))
                out = uploaded.pop(ci)
                if torch is None:
                    yield out
                else:
                    d, ev = out
                    ev.synchronize()      # the library launches on its own stream: hand over completed uploads only
                    yield d
        finally:
            if self.processes > 0:
                pool.terminate()
                pool.join()
                if shm is not None:
                    shm.close()
                    shm.unlink()
            else:
                pool.shutdown(wait=True, cancel_futures=True)
