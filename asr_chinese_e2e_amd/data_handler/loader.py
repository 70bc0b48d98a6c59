"""Waveform dataset + length-bucketed loader with the feature front end on the GPU
(SURVEY.md 8(f) rank 2).

The reference's input pipeline (data/data_loader/ai_shell_1.py:12-104) computes the features on
the CPU per utterance (torchaudio), pads whole batches of FEATURES on the host, is unshuffled,
unbucketed, single-process, and moves the batch to the GPU inside collate.  Here the host only
decodes PCM and pads WAVEFORMS into a pinned buffer; the copy runs on its own stream, log-mel /
normalisation / SpecAugment / frame stacking run on the GPU (data_handler.processor.AudioParser),
and the next batch is prepared while the model trains on the current one.  Batches are drawn
from length buckets so that padding stays small.

The yielded Pack has the reference's batch contract (collat, ai_shell_1.py:75-88): wave (B, T, F),
wave_len, tgt_for_input, tgt_for_metric (0-padded, int64), tgt_len.
"""
import contextlib
import random
import threading
import wave as wave_module
import weakref

import numpy as np
import torch

from ..Utils import Pack
from .padder import Padder
from .processor import AudioParser


def load_wav(path):
    """16-bit PCM WAV -> (float32 mono waveform in [-1, 1), sample_rate); channels are averaged
    (loader.py:5-17 of the reference did the same through torchaudio with normalization=True)."""
    with wave_module.open(path, "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM is supported (sample width {f.getsampwidth()})")
        sr, ch, n = f.getframerate(), f.getnchannels(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2").astype(np.float32) / 32768.0
    if ch > 1:
        pcm = pcm.reshape(-1, ch).mean(axis=1)
    return pcm, sr


class WaveDataset:
    """items: list of (waveform, text) where waveform is a 1-D float array / tensor or a path to a
    16-bit WAV file, and text is a string (converted by `vocab.convert_str(..., use_bos=False,
    use_eos=False)` as ai_shell_1.py:53-54) or a list of ids."""

    def __init__(self, items, vocab=None, sample_rate=16000):
        self.items, self.vocab, self.sample_rate = items, vocab, sample_rate
        self._len = [None] * len(items)

    def __len__(self):
        return len(self.items)

    def wave(self, i):
        w = self.items[i][0]
        if isinstance(w, str):
            w, sr = load_wav(w)
            if sr != self.sample_rate:
                raise ValueError(f"{self.items[i][0]}: sample rate {sr}, expected {self.sample_rate}")
        return np.asarray(w, dtype=np.float32).reshape(-1)

    def wave_into(self, i, row):
        """Utterance i decoded straight into `row` (a float32 numpy view of the loader's pinned staging buffer), the rest of the row zeroed;
        returns the sample count.  16-bit PCM goes int16 -> float32 in one pass (no intermediate array); numpy releases the GIL for it."""
        w = self.items[i][0]
        if isinstance(w, str):
            with wave_module.open(w, "rb") as f:
                if f.getsampwidth() != 2:
                    raise ValueError(f"{w}: only 16-bit PCM is supported (sample width {f.getsampwidth()})")
                sr, ch, n = f.getframerate(), f.getnchannels(), f.getnframes()
                if sr != self.sample_rate:
                    raise ValueError(f"{w}: sample rate {sr}, expected {self.sample_rate}")
                pcm = np.frombuffer(f.readframes(n), dtype="<i2")
            # the sample count is what the file actually holds (a data chunk shorter than its header says is a truncated recording, not an
            # error), never more than the row the loader sized from the header
            n = min(pcm.size // ch, row.size)
            if ch > 1:      # same arithmetic as load_wav: scale in float32, then average the channels
                row[:n] = (pcm[:n * ch].astype(np.float32) / np.float32(32768.0)).reshape(-1, ch).mean(axis=1)
            else:
                np.multiply(pcm[:n], np.float32(1.0 / 32768.0), out=row[:n], casting="unsafe")
        else:
            a = np.asarray(w, dtype=np.float32).reshape(-1)
            n = a.size
            row[:n] = a
        row[n:] = 0.0
        return n

    def num_samples(self, i):
        if self._len[i] is None:
            w = self.items[i][0]
            if isinstance(w, str):
                with wave_module.open(w, "rb") as f:
                    self._len[i] = f.getnframes()
            else:
                self._len[i] = int(np.asarray(w).size)
        return self._len[i]

    def ids(self, i):
        t = self.items[i][1]
        if isinstance(t, str):
            return self.vocab.convert_str(t, use_bos=False, use_eos=False)
        return [int(x) for x in t]


def bucket_batches(lengths, batch_size, bucket_size=None, shuffle=True, drop_last=False, rng=None):
    """Index batches drawn from length buckets: indices sorted by length are cut into buckets of
    `bucket_size` (default 8 batches), each bucket is shuffled and cut into batches, the batches
    are shuffled.  Every index appears exactly once (or is dropped with a short last batch)."""
    rng = rng or random
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    bucket_size = bucket_size or 8 * batch_size
    batches = []
    for s in range(0, len(order), bucket_size):
        b = order[s:s + bucket_size]
        if shuffle:
            rng.shuffle(b)
        for t in range(0, len(b), batch_size):
            batches.append(b[t:t + batch_size])
    if drop_last:
        batches = [b for b in batches if len(b) == batch_size]
    if shuffle:
        rng.shuffle(batches)
    return batches


def shard_batches(batches, rank, world):
    """Data-parallel shard of a batch list that every rank builds identically: the ragged tail of len % world is
    dropped (every rank must take the same number of steps) and rank r takes batches[r::world]."""
    if world <= 1:
        return batches
    return batches[: len(batches) // world * world][rank::world]


_LOADERS = weakref.WeakSet()      # loaders of this process (paused() holds every one's gate)


@contextlib.contextmanager
def paused():
    """No loader's helper thread touches the GPU runtime inside this block: each helper takes its loader's gate around the preparation of a
    batch (pinned allocations, copies, front-end kernels, event calls), and this context holds all the gates.  graph.GraphedStep wraps its
    warm-up and capture in it: under torch.cuda.graph's default capture mode an allocation or an event synchronisation on ANOTHER thread
    invalidates the capture (round-4 ADVICE).  Work a helper has already queued on its stream keeps running - only API calls matter."""
    gates = [l._gate for l in list(_LOADERS)]
    for g in gates:
        g.acquire()
    try:
        yield
    finally:
        for g in reversed(gates):
            g.release()


class BucketedWaveLoader:
    """Iterating yields Packs on `device`; batches are prepared ahead on the loader's stream by a helper thread.
    A yielded Pack (and every tensor in it) is valid until the NEXT one is requested: its memory belongs to a staging slot that the helper
    reuses SLOTS batches later, ordered behind an event the consumer's stream records when it asks for the next batch (copy what must live
    longer)."""

    def __init__(self, dataset, batch_size, parser=None, augment=False, shuffle=True, drop_last=False, seed=0, bucket_size=None,
                 device="cuda", dtype=torch.bfloat16, rank=0, world=1):
        """rank / world: data-parallel sharding.  Every rank draws the SAME batch list (same seed), keeps only the full
        batches when world > 1 (dist.DataParallel normalises by world x local batch and every rank must take the same
        number of steps), drops the ragged tail of len(batches) % world and takes batches[rank::world]."""
        self.ds, self.batch_size, self.augment, self.shuffle, self.drop_last = dataset, batch_size, augment, shuffle, drop_last
        self.rank, self.world = int(rank), int(world)
        assert 0 <= self.rank < self.world
        if self.world > 1:
            self.drop_last = True
        self.device, self.dtype = torch.device(device), dtype
        if self.device.type != "cuda":
            raise RuntimeError("the feature front end runs on the GPU only (no CPU fallback)")
        self.parser = parser or AudioParser(device=self.device)
        self.rng = random.Random(seed)          # batch order AND SpecAugment masks (the reference uses the global `random`)
        self.bucket_size = bucket_size
        self.stream = self._copy_stream()
        self.lengths = [dataset.num_samples(i) for i in range(len(dataset))]
        self._gate = threading.Lock()      # held by the helper thread around _prepare; paused() takes it
        _LOADERS.add(self)

    def _copy_stream(self):
        """The stream of the host-to-device copies and the feature front end of the NEXT batch: ONE per device and process, on a hardware
        queue other than the training stream's and the engine's weight-gradient / auxiliary streams' - whichever of the two is built first
        (engine.shared_stream keeps the registry: a stream chosen later avoids every stream chosen before it)."""
        from .. import engine as E
        return E.shared_stream(self.device, "loader")

    def __len__(self):
        n = len(self.ds)
        if self.world > 1:      # full batches only (per bucket), the same count on every rank
            return len(self._batches(random.Random(0)))
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _batches(self, rng):
        batches = bucket_batches(self.lengths, self.batch_size, self.bucket_size, self.shuffle, self.drop_last, rng)
        return shard_batches(batches, self.rank, self.world)

    # ---- staging: pinned host buffers that are REUSED (a fresh pageable tensor + pin_memory() + 32 tensor-slice assignments + six small
    # pageable host-to-device copies cost 25 ms of host time per batch of 32 x 5 s - the training step takes 3 ms), filled through
    # numpy views (plain memcpy, the GIL released), one buffer for the samples and ONE for every integer of the batch
    SLOTS = 4

    def _slot(self, k, n_wave, n_meta):
        """Staging slot k, free for a new batch.  A slot owns the pinned host buffers AND the device tensors of the batch it staged last
        (`keep`): those go back to the loader stream's pool only here, SLOTS batches later, after the loader's stream has been made to wait
        for `consumed` - the event the consumer's stream recorded when it asked for the batch after that one.  (Round 4 marked every tensor
        of a pack with Tensor.record_stream instead: the mechanism that made the caching allocator regrow by hipMalloc in the engine.)"""
        slots = self.__dict__.setdefault("_slots", [None] * self.SLOTS)
        s = slots[k]
        if s is not None:
            s["copied"].synchronize()      # the copy out of this slot (SLOTS batches ago) has run: the host may overwrite the pinned buffers
            if s["handed"]:
                self.stream.wait_event(s["consumed"])      # recorded before the consumer took the batch after this slot's (see __iter__)
            s["keep"], s["handed"] = None, False
        if s is None or s["wave"].numel() < n_wave or s["meta"].numel() < n_meta:
            cap_w = max(n_wave, s["wave"].numel() if s else 0) * 5 // 4
            cap_m = max(n_meta, s["meta"].numel() if s else 0) * 2
            s = dict(wave=torch.empty(cap_w, dtype=torch.float32, pin_memory=True), meta=torch.empty(cap_m, dtype=torch.int32, pin_memory=True),
                     copied=torch.cuda.Event(), consumed=torch.cuda.Event(), keep=None, handed=False)
            s["wave_np"], s["meta_np"] = s["wave"].numpy(), s["meta"].numpy()
            slots[k] = s
        return s

    def _prepare(self, idx, k=0):
        tgt = [self.ds.ids(i) for i in idx]
        B = len(idx)
        into = getattr(self.ds, "wave_into", None)
        waves = None if into is not None else [self.ds.wave(i) for i in idx]
        smax = max(self.lengths[i] for i in idx) if waves is None else max(w.size for w in waves)
        lmax = max(1, max(len(t) for t in tgt))
        slot = self._slot(k % self.SLOTS, B * smax, 2 * B + B * lmax)
        buf = slot["wave_np"][:B * smax].reshape(B, smax)
        meta = slot["meta_np"][:2 * B + B * lmax]
        tg = meta[2 * B:].reshape(B, lmax)
        items = getattr(self.ds, "items", None)
        if waves is None and items is not None and any(isinstance(items[i][0], str) for i in idx):
            # files: read / decode the rows in parallel (file reads and numpy loops run without the GIL); 3.4 - 3.8 ms/step against 4 - 6 serially
            pool = self.__dict__.get("_pool")
            if pool is None:
                import concurrent.futures
                pool = self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="asr-decode")
            sizes = list(pool.map(lambda r: into(idx[r], buf[r]), range(B)))
        elif waves is None:      # arrays in memory: 32 memcpys are cheaper than handing them to a pool (3.24 vs 3.35 ms/step)
            sizes = [into(idx[r], buf[r]) for r in range(B)]
        else:
            sizes = []
            for r, w in enumerate(waves):
                buf[r, :w.size] = w
                buf[r, w.size:] = 0.0
                sizes.append(w.size)
        for r, t in enumerate(tgt):
            meta[r], meta[B + r] = sizes[r], len(t)
            tg[r, :len(t)] = t
            tg[r, len(t):] = 0
        with torch.cuda.stream(self.stream):
            dev_wav = slot["wave"][:B * smax].view(B, smax).to(self.device, non_blocking=True)
            dev_meta = slot["meta"][:meta.size].to(self.device, non_blocking=True)
            slot["copied"].record()
            feat, feat_len = self.parser.parse_batch(dev_wav, dev_meta[:B], self.dtype, augment=self.augment, rng=self.rng)
            tgt_dev = dev_meta[2 * B:].view(B, lmax).long()
            pack = Pack()
            pack.add(wave=feat, wave_len=feat_len.long(), tgt_for_input=tgt_dev, tgt_for_metric=tgt_dev.clone(), tgt_len=dev_meta[B:2 * B].long())
            done = torch.cuda.Event()
            done.record()
        slot["keep"] = (dev_wav, dev_meta, feat, feat_len) + tuple(v for v in pack.values() if torch.is_tensor(v))
        return pack, done, slot

    PREFETCH = 2      # batches prepared ahead by the helper thread

    def __iter__(self):
        """Batches are prepared by a helper thread, PREFETCH ahead: decoding / padding into the pinned slot (memcpy: no GIL), the two
        host-to-device copies and the feature kernels on the loader's stream.  The consumer's stream waits for the batch's event."""
        import queue
        from .. import kernels as K
        batches = self._batches(self.rng)
        q = queue.Queue(maxsize=self.PREFETCH)
        stop = threading.Event()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()      # the consumer's device

        def hand_over(item):      # False: the consumer is gone
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            K._TLS.own_stream = True      # never follow the training thread's stream override (kernels._stream)
            try:
                torch.cuda.set_device(dev_index)
                for k, idx in enumerate(batches):
                    with self._gate:      # paused() (a hipGraph capture on the consumer thread) keeps this thread off the GPU runtime
                        item = self._prepare(idx, k)
                    if not hand_over(item):
                        return
                hand_over(None)
            except BaseException as e:      # noqa: BLE001 - handed to the consumer
                hand_over(e)

        th = threading.Thread(target=work, name="asr-loader", daemon=True)
        th.start()
        last = None      # slot of the batch the consumer is working on
        try:
            while True:
                if last is not None:
                    # the consumer is back for the next batch: everything it queued on its stream with the previous one is in front of this
                    # event, and the helper makes the loader's stream wait for it before that slot's device memory is reused (SLOTS batches
                    # later; the queue's depth guarantees the record happens before the helper gets there, see _slot)
                    last["consumed"].record(torch.cuda.current_stream())
                    last["handed"] = True
                    last = None
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                pack, done, last = item
                torch.cuda.current_stream().wait_event(done)
                yield pack
        finally:
            stop.set()
            th.join(timeout=5.0)
            if last is not None:      # the consumer left in the middle of an epoch
                last["consumed"].record(torch.cuda.current_stream())
                last["handed"] = True

def build_dataloader(collector_path, vocab, batch_size, part="test", use_cuda=True, sample_rate=16000, window_size=400, n_mels=40,
                     augment=False, predump=False, use_old=False, lfr_m=4, lfr_n=3, dtype=torch.bfloat16, shuffle=None, seed=0,
                     rank=0, world=1):
    """build_dataloader of the reference (data/data_loader/ai_shell_1.py:91-104), same arguments: reads the manifest
    `<collector_path>_<part>.json` written by the reference's collector (one JSON object {"wave": path, "tgt": text}
    per line, data_collector/ai_shell_1.py:73-79) and returns an iterable of Packs.  The reference computes features
    on the CPU per utterance and can cache them as .t files (predump / use_old); here they are computed on the GPU
    per batch, so both flags are accepted and ignored.  drop_last=True as in the reference (:103)."""
    import json
    if not use_cuda:
        raise RuntimeError("the feature front end runs on the GPU only (no CPU fallback)")
    if window_size != 400:
        raise ValueError("the log-mel kernel is built for the reference's window of 400 samples (data_config.py:13)")
    items = []
    with open(collector_path + "_" + part + ".json", encoding="utf-8") as reader:
        for line in reader:
            if line.strip():
                rec = json.loads(line)
                items.append((rec["wave"], rec["tgt"]))
    ds = WaveDataset(items, vocab, sample_rate=sample_rate)
    parser = AudioParser(sample_rate=sample_rate, n_mels=n_mels, lfr_m=lfr_m, lfr_n=lfr_n, device="cuda")
    return BucketedWaveLoader(ds, batch_size, parser=parser, augment=augment, shuffle=(part == "train") if shuffle is None else shuffle,
                              drop_last=True, seed=seed, dtype=dtype, rank=rank, world=world)
