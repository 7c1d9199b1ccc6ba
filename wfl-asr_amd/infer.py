"""The inference surface of WFL-ASR on the MI355X path: same functions, arguments, files and `.lab` output as
/root/reference/infer.py, with the per-file model rebuild and the B=1 forwards replaced by one resident model and a
batched loop over 30 s work items.

  infer_audio / infer_folder   signatures of infer.py:186-191 / 330-332
  Labeler                      the batched loop: one model load (the reference reloads per file, infer.py:205-208),
                               every clip <= 30 s and every 30 s chunk of a longer file is one batch row; rows of many
                               files share a forward; `lang_id=None` averages logits/offsets over all languages
                               inside the library (encoder runs once; infer.py:146-156, 266-276)
  deviations (documented in DESIGN.md): no `.wfl_cache` (infer.py:223-229), `--sample/--top-k/--top-p/--temperature`
  are validated but have no effect (their results are overwritten in the reference too, infer.py:283-297), a single
  file with `-o .` writes `<stem>.lab` instead of overwriting the input WAV (infer.py:410-411, utils.py:77), and
  `--device cpu` is refused (no CPU path).
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch
import yaml

from . import audio as A
from . import native_post as npost
from . import postprocess as pp
from .tagger import BIOPhonemeTagger, raise_on_status

frame_duration = pp.FRAME_DURATION
MAX_SEGMENT_DURATION = pp.MAX_SEGMENT_DURATION
CHUNK_SAMPLES = int(MAX_SEGMENT_DURATION * 16000)


def load_config(config_path="config.yaml"):
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def pick_device(device="cuda") -> torch.device:
    """The HIP device this process labels on.  An explicit index wins; a bare "cuda" means cuda:LOCAL_RANK under a
    one-process-per-GPU launcher (torchrun sets LOCAL_RANK and WORLD_SIZE) and the current device otherwise."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("this build labels on MI355X only: --device must be cuda[:N] (there is no CPU path)")
    if dev.index is not None:
        return dev
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "LOCAL_RANK" in os.environ:
        return torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    return torch.device("cuda", torch.cuda.current_device())


class Labeler:
    """Model + sidecar files loaded once; `label_files` runs the batched hot loop."""

    def __init__(self, config_path, checkpoint_path, device="cuda", batch_size=None, use_graph=False):
        self.config = load_config(config_path) if isinstance(config_path, (str, os.PathLike)) else config_path
        if torch.device(device).type == "cuda" and not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible")
        self.device = pick_device(device)
        torch.cuda.set_device(self.device)       # weights, workspaces, streams and pinned staging all belong to this device
        self.sr = int(self.config["data"]["sample_rate"])
        if self.sr != 16000:
            raise ValueError("both encoders are 16 kHz models (config data.sample_rate must be 16000)")
        save_dir = self.config["output"]["save_dir"]
        self.labels = pp.load_phoneme_list(os.path.join(save_dir, "phonemes.txt"))
        langs_path = os.path.join(save_dir, "langs.txt")
        self.lang2id = pp.load_langs(langs_path) if os.path.exists(langs_path) else {}
        mm_path = os.path.join(save_dir, "phoneme_merge_map.json")
        self.merge_map = pp.load_phoneme_merge_map(mm_path) if os.path.exists(mm_path) else None
        self.model = BIOPhonemeTagger(self.config, self.labels, device=self.device)
        if isinstance(checkpoint_path, dict):
            state_dict = checkpoint_path
        else:
            state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(state_dict)
        self.model.to(self.device).eval()
        if self.lang2id:
            # `lang_id=None` averages over the ids langs.txt lists, in file order (infer.py:147-156, 266-276)
            self.model.set_average_languages(list(self.lang2id.values()))
        if batch_size is None:
            # rows per forward.  A BiLSTM's recurrence costs the same ~1.2 us per time step for 16 clips as for 64 (clips run in
            # groups of 16 on their own workgroups), so wide batches amortise it: default head, 64 rows 107 k audio-s/s, 16 rows 87 k
            # (DESIGN.md section 5).  A clip's tags do not depend on what shares its batch (bit-exact batch invariance).
            batch_size = int(os.environ.get("WFL_BATCH_SIZE", "64" if self.model.head_cfg["enable_bilstm"] else "16"))
        self.batch_size = int(batch_size)
        self.use_graph = bool(use_graph)
        self._pinned = None
        self.n_inflight = self.model.batches_in_flight()   # batches on the GPU at once: stream + workspace slot each
        self._table = npost.LabelTable(self.labels)          # native BIO decode works on ids (csrc/hostpost.hip)
        self._streams = None

    # ------------------------------------------------------------------ the batched loop
    def _forward_items(self, items, lang_id, threshold):
        """items: list of float32 arrays (<= 480000 samples each) -> list of (ids[T], offsets[T,2]) numpy."""
        if self.model.encoder_type != "whisper":
            return self._forward_items_by_length(items, lang_id, threshold)
        out = [None] * len(items)
        Bs = self.batch_size
        L = CHUNK_SAMPLES

        def fill(k, host):
            chunk = items[k * Bs:(k + 1) * Bs]
            lens = np.zeros(Bs, dtype=np.int32)
            rows = host.numpy()                  # (a numpy row assignment is a plain memcpy, 0.2 ms per 30 s item; the same through
            for i, x in enumerate(chunk):        #  torch's indexing was measured at 1-18 ms)
                n = min(len(x), L)
                rows[i, :n] = x[:n]
                lens[i] = n
            return lens, len(chunk)

        def take(k, n, ids, offs):
            for i in range(n):
                out[k * Bs + i] = (ids[i].copy(), offs[i].copy())

        self._run_batches((len(items) + Bs - 1) // Bs, fill, take, lang_id, threshold)
        return out

    def _run_batches(self, n_batches, fill, take, lang_id, threshold, pinned_in=None, upload=None):
        """The pipelined hot loop.  NS = `n_inflight` batches in flight on the GPU (two, or three for a BiLSTM behind a small
        encoder -- tagger.batches_in_flight; stream and workspace slot k % NS, own pinned output buffer), one more being filled:
        `fill(k, pinned_rows) -> (lens, rows used)` runs on a worker thread one batch ahead of the launches (the native loader
        releases the GIL), into a ring of NS + 1 pinned input buffers, each guarded by the event of its last host-to-device
        copy; the main thread launches batch k and unpacks batch k - NS (`take(k, rows, ids[B, T], offsets[B, T, 2])`, views
        into pinned memory: copy what you keep).  With everything on one thread (round 2's first half)
        the end-to-end rate of a folder of 30 s files was 82 k audio-s/s against 123 k for batches already resident."""
        from concurrent.futures import ThreadPoolExecutor
        Bs = self.batch_size
        L = CHUNK_SAMPLES
        T = self.model.num_frames(L)
        NS = self.n_inflight
        NI = NS + 1
        # pinned_in / upload: another staging format (the GPU ingest path: int16 PCM rows, resampled on the device -- _label_resampled);
        # default: float32 waveform rows copied as they are
        if pinned_in is None and (self._pinned is None or len(self._pinned) != NI or self._pinned[0].shape[0] != Bs):
            self._pinned = [torch.zeros(Bs, L, dtype=torch.float32).pin_memory() for _ in range(NI)]
        if getattr(self, "_pinned_out", None) is None or len(self._pinned_out) != NS or self._pinned_out[0].numel() != Bs * T * 4 + 1:
            self._pinned_out = [torch.empty(Bs * T * 4 + 1, dtype=torch.int32).pin_memory() for _ in range(NS)]
        pin = pinned_in if pinned_in is not None else self._pinned
        assert len(pin) == NI
        self._ensure_streams()
        use_pipe = not self.use_graph
        pending = [None] * NS
        copied = [None] * NI                               # event behind the last H2D copy out of input buffer i

        def finish(slot):
            job = pending[slot]
            if job is None:
                return
            k, n, ev = job
            ev.synchronize()
            blob = self._pinned_out[slot].numpy()
            nn = Bs * T
            raise_on_status(int(blob[4 * nn]))             # the forward's device-side error word rides behind the tags
            take(k, n, blob[0:nn].reshape(Bs, T), blob[2 * nn:4 * nn].view(np.float32).reshape(Bs, T, 2))
            pending[slot] = None

        def fill_job(k):
            ib = k % NI
            if copied[ib] is not None:
                copied[ib].synchronize()                   # (NS + 1 batches back: long done)
            return fill(k, pin[ib])

        if n_batches <= 0:
            return
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(fill_job, 0)
            for k in range(n_batches):
                lens, n = fut.result()
                if k + 1 < n_batches:
                    fut = pool.submit(fill_job, k + 1)
                slot = k % NS if use_pipe else 0
                finish(slot)                               # the output buffer and workspace of this slot are free again
                host = pin[k % NI]
                stream = self._streams[slot] if use_pipe else torch.cuda.current_stream(self.device)
                with torch.cuda.stream(stream):
                    if upload is None:
                        dev_wav = host.to(self.device, non_blocking=True)
                    else:
                        dev_wav, lens = upload(host, lens, slot, stream)
                    ev_in = torch.cuda.Event()
                    ev_in.record(stream)
                    copied[k % NI] = ev_in
                    res = self.model.label(dev_wav, None if lang_id is None else [lang_id] * Bs, threshold=threshold, lens=lens,
                                           average_languages=lang_id is None, graph=self.use_graph, slot=slot)
                    self._pinned_out[slot].copy_(res.packed, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                pending[slot] = (k, n, ev)
            for k in range(max(0, n_batches - NS), n_batches):   # what is still in flight, oldest first
                finish(k % NS if use_pipe else 0)

    def _ensure_streams(self):
        if self._streams is None or len(self._streams) != self.n_inflight:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(self.n_inflight)]

    def _forward_items_by_length(self, items, lang_id, threshold):
        """WavLM and the mel front-end: the frame count follows the clip length and the reference never pads their input (padding
        would change the waveform / GroupNorm statistics, the unmasked attention and where the backward LSTM starts).  A batch may
        still hold clips of different lengths: the library takes per-clip sample counts (`lens`) and carries every clip's own frame
        count through the whole forward, so each row comes out as if labelled alone (csrc/model.hip, Runner::clipT).  Clips are
        sorted by length so that a batch wastes little on its shorter rows; `WFL_RAGGED=0` goes back to one length per batch.
        Same pipeline as `_run_batches`: `n_inflight` batches in flight on their streams / workspace slots, pinned staging both ways, the
        status word read with the tags.  (Two WavLM-base forwards in flight used to disturb each other now and then: a packed-f32
        instruction form in the group-norm conv0 kernel that is unsafe beside another wave's MFMAs -- DESIGN.md section 7; the form is
        gone and the build refuses it; `test_two_wavlm_forwards_in_flight_do_not_disturb_each_other` guards the loop.)"""
        out = [None] * len(items)
        Bs = self.batch_size
        ragged = os.environ.get("WFL_RAGGED", "1") != "0"
        frames = [self.model.num_frames(len(x)) for x in items]
        for i, t in enumerate(frames):
            if t <= 0:
                raise ValueError(f"clip {i} is too short for the {self.model.encoder_type} front-end ({len(items[i])} samples)")
        if ragged:
            order = sorted(range(len(items)), key=lambda i: -len(items[i]))
            jobs = [order[s:s + Bs] for s in range(0, len(order), Bs)]
        else:
            by_len = {}
            for i, x in enumerate(items):
                by_len.setdefault(len(x), []).append(i)
            jobs = [idxs[s:s + Bs] for n, idxs in by_len.items() for s in range(0, len(idxs), Bs)]
        self._ensure_streams()
        NS = self.n_inflight
        NI = NS + 1
        if not jobs:
            return out
        # pinned staging sized once for the largest batch (round 2 grew the buffers lazily inside the loop): NS + 1 input buffers, so that
        # a worker thread fills batch k + 1 while the main thread launches batch k (the same pipeline as _run_batches), NS output buffers
        geom = []
        for sel in jobs:
            n = max(len(items[i]) for i in sel)
            geom.append((n, self.model.num_frames(n), all(len(items[i]) == n for i in sel)))
        need_in = max(len(sel) * g[0] for sel, g in zip(jobs, geom))
        need_out = max(len(sel) * g[1] * 4 + 1 for sel, g in zip(jobs, geom))
        cache = getattr(self, "_pinned_ragged", None)
        if cache is None or len(cache[0]) != NI or cache[0][0].numel() < need_in or len(cache[1]) != NS or cache[1][0].numel() < need_out:
            cache = ([torch.empty(need_in, dtype=torch.float32).pin_memory() for _ in range(NI)],
                     [torch.empty(need_out, dtype=torch.int32).pin_memory() for _ in range(NS)])
            self._pinned_ragged = cache
        pin_in, pin_out = cache
        pending = [None] * NS
        copied = [None] * NI                               # event behind the last H2D copy out of input buffer i

        def finish(slot):
            job = pending[slot]
            if job is None:
                return
            sel, T, ev = job
            ev.synchronize()
            nn = len(sel) * T
            blob = pin_out[slot].numpy()
            raise_on_status(int(blob[4 * nn]))
            ids = blob[0:nn].reshape(len(sel), T)
            offs = blob[2 * nn:4 * nn].view(np.float32).reshape(len(sel), T, 2)
            for j, i in enumerate(sel):
                out[i] = (ids[j, :frames[i]].copy(), offs[j, :frames[i]].copy())
            pending[slot] = None

        def fill_job(k):
            sel = jobs[k]
            n, T, same = geom[k]
            ib = k % NI
            if copied[ib] is not None:
                copied[ib].synchronize()                   # (NS + 1 batches back: long done)
            host = pin_in[ib][:len(sel) * n].view(len(sel), n)
            rows = host.numpy()
            for j, i in enumerate(sel):
                m = len(items[i])
                rows[j, :m] = items[i]
                if m < n:
                    rows[j, m:] = 0.0                      # (behind a shorter clip: never read -- `lens` -- but keep the buffer defined)
            return host, (None if same else np.array([len(items[i]) for i in sel], np.int32))

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(fill_job, 0)
            for k, sel in enumerate(jobs):
                host, lens = fut.result()
                if k + 1 < len(jobs):
                    fut = pool.submit(fill_job, k + 1)
                slot = k % NS
                finish(slot)
                n, T, same = geom[k]
                need_o = len(sel) * T * 4 + 1
                with torch.cuda.stream(self._streams[slot]):
                    wav = host.to(self.device, non_blocking=True)
                    ev_in = torch.cuda.Event()
                    ev_in.record(self._streams[slot])
                    copied[k % NI] = ev_in
                    res = self.model.label(wav, None if lang_id is None else [lang_id] * len(sel), threshold=threshold, lens=lens,
                                           average_languages=lang_id is None, slot=slot)
                    pin_out[slot][:need_o].copy_(res.packed, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._streams[slot])
                pending[slot] = (sel, T, ev)
            for k in range(max(0, len(jobs) - NS), len(jobs)):
                finish(k % NS)
        return out

    def _lang_name(self, lang_id):
        if lang_id is None:
            return None
        for n, i in self.lang2id.items():
            if i == lang_id:
                return n
        return None

    def _names_for(self, lang_name):
        """phoneme index -> (index into unique output names, names): the merge-map back-mapping (utils.py:206-211) applied to
        the phoneme table once instead of to every segment; phonemes that map to the same string share an index, so the
        equal-neighbour merge compares exactly what the reference's string comparison compares."""
        key = lang_name if (self.merge_map and lang_name) else None
        cache = getattr(self, "_names_cache", None)
        if cache is None:
            cache = self._names_cache = {}
        if key not in cache:
            mapped = [pp.canonical_to_lang(ph, key, self.merge_map) if key else ph for ph in self._table.names]
            uniq, remap = [], np.empty(max(len(mapped), 1), np.int32)
            index = {}
            for i, nm in enumerate(mapped):
                if nm not in index:
                    index[nm] = len(uniq)
                    uniq.append(nm)
                remap[i] = index[nm]
            cache[key] = (remap, uniq)
        return cache[key]

    def _segments_of_item(self, ids, offsets, lang_name):
        """suppressed ids -> median filter -> BIO decode -> merge-map back-mapping (infer.py:164-179, 293-307), in native
        code; returns (start[n] f64, end[n] f64, name index[n]) arrays."""
        mf = int(self.config["postprocess"]["median_filter"])
        s, e, ph = npost.decode_bio_ids(ids, self._table, frame_duration, offsets, median=mf)
        remap, _ = self._names_for(lang_name)
        return s, e, remap[ph] if ph.size else ph

    def _label_fast(self, audio_paths, lang_id, threshold, lang_name):
        """Files the native loader takes whole (16 kHz, <= 30 s, <= 2 channels, PCM / float WAV): decoded, normalised and
        converted by worker threads straight into the pinned batch rows (audio.load_wavs_into), one file per row; a finished
        batch's tags go to a second worker thread for the native median filter + BIO decode + segment merge while the next
        batches run.  Returns {file index: [(start_s, end_s, phoneme)]} (before forced alignment); every other file is left to the
        general path."""
        from concurrent.futures import ThreadPoolExecutor
        Bs, L = self.batch_size, CHUNK_SAMPLES
        done = {}
        meta = {}
        self._names_for(lang_name)                       # (build the cache on this thread)
        post = ThreadPoolExecutor(max_workers=1)
        futures = []
        threads = max(1, min(16, (os.cpu_count() or 1)))

        def fill(k, host):
            sel = list(range(k * Bs, min((k + 1) * Bs, len(audio_paths))))
            ns, srs, st = A.load_wavs_into([audio_paths[i] for i in sel], host, L, threads)
            lens = np.zeros(Bs, dtype=np.int32)
            ok = []
            for r, fi in enumerate(sel):
                if st[r] == 0 and srs[r] == self.sr and ns[r] > 0:
                    lens[r] = ns[r]
                    ok.append((r, fi, int(ns[r])))
            meta[k] = ok
            return lens, len(sel)

        mode = self.config["postprocess"]["merge_segments"]
        names = self._names_for(lang_name)[1]

        def segments(rows):
            res = []
            for fi, ids, offs in rows:
                s, e, ph = self._segments_of_item(ids, offs, lang_name)
                ph = ph.astype(np.int32)
                if mode != "none" and s.size:
                    s, e, ph = npost.merge_segments(s, e, ph, mode)
                res.append((fi, npost.to_tuples(s, e, ph, names)))
            return res

        def take(k, n, ids, offs):
            futures.append(post.submit(segments, [(fi, ids[r].copy(), offs[r].copy()) for r, fi, ns in meta.pop(k)]))

        try:
            self._run_batches((len(audio_paths) + Bs - 1) // Bs, fill, take, lang_id, threshold)
            for f in futures:
                for fi, seg in f.result():
                    done[fi] = seg
        finally:
            post.shutdown(wait=True)
        return done

    def _label_resampled(self, audio_paths, src_sr, lang_id, threshold, lang_name):
        """Files of ONE sample rate other than the model's, 16-bit PCM, at most 30 s: their samples go to the GPU as they are (the native
        reader copies the file's PCM bytes into pinned int16 rows) and csrc/resample.hip decodes, resamples (float64 sinc, torchaudio's
        published algorithm: infer.py:217-220) and peak-normalises (infer.py:234-235) them into the batch the forward reads -- the host's
        cores no longer bound a 44.1 kHz folder (48 k audio-s/s with the host resampler).  Same pipeline, post-
        processing and result as _label_fast."""
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor

        from . import _lib
        lib = _lib.load()
        Bs, L = self.batch_size, CHUNK_SAMPLES
        NI = self.n_inflight + 1
        cap_in = 2 * (int(math.ceil(L * src_sr / self.sr)) + 2)          # interleaved int16 values of a 30 s two-channel file
        key = (src_sr, Bs)
        cache = getattr(self, "_pinned_pcm", None)
        if cache is None or cache[0] != key or len(cache[1]) != NI:
            cache = (key, [torch.zeros(Bs, cap_in, dtype=torch.int16).pin_memory() for _ in range(NI)])
            self._pinned_pcm = cache
        pin = cache[1]
        ws_bytes = int(lib.wfl_resample_workspace_bytes(Bs, L))
        dev_out, dev_ws = {}, {}
        meta, done, futures = {}, {}, []
        self._names_for(lang_name)
        post = ThreadPoolExecutor(max_workers=1)
        threads = max(1, min(16, (os.cpu_count() or 1)))

        def fill(k, host):
            sel = list(range(k * Bs, min((k + 1) * Bs, len(audio_paths))))
            nf, ch, srs, st = A.read_pcm16_into([audio_paths[i] for i in sel], host, cap_in, threads)
            frames = np.zeros(Bs, np.int32)
            chans = np.ones(Bs, np.int32)
            ok = []
            for r, fi in enumerate(sel):
                if st[r] == 0 and srs[r] == src_sr and nf[r] > 0 and int(math.ceil(int(nf[r]) * self.sr / src_sr)) <= L:
                    frames[r], chans[r] = nf[r], ch[r]
                    ok.append((r, fi))
            meta[k] = ok
            return (frames, chans), len(sel)

        def upload(host, fc, slot, stream):
            frames, chans = fc
            if slot not in dev_out:
                dev_out[slot] = torch.zeros(Bs, L, dtype=torch.float32, device=self.device)
                dev_ws[slot] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            d_pcm = host.to(self.device, non_blocking=True)
            d_meta = torch.from_numpy(np.stack([frames, chans])).to(self.device)
            rc = lib.wfl_resample_pcm16(C.c_void_p(d_pcm.data_ptr()), cap_in, C.c_void_p(d_meta[0].data_ptr()), C.c_void_p(d_meta[1].data_ptr()),
                                        Bs, int(src_sr), int(self.sr), C.c_void_p(dev_out[slot].data_ptr()), L, L,
                                        C.c_void_p(dev_ws[slot].data_ptr()), ws_bytes, C.c_void_p(stream.cuda_stream))
            _lib.check(rc, "wfl_resample_pcm16")
            d_pcm.record_stream(stream)
            d_meta.record_stream(stream)
            lens = np.minimum(np.ceil(frames.astype(np.float64) * self.sr / src_sr), L).astype(np.int32)
            return dev_out[slot], lens

        mode = self.config["postprocess"]["merge_segments"]
        names = self._names_for(lang_name)[1]

        def segments(rows):
            res = []
            for fi, ids, offs in rows:
                s, e, ph = self._segments_of_item(ids, offs, lang_name)
                ph = ph.astype(np.int32)
                if mode != "none" and s.size:
                    s, e, ph = npost.merge_segments(s, e, ph, mode)
                res.append((fi, npost.to_tuples(s, e, ph, names)))
            return res

        def take(k, n, ids, offs):
            futures.append(post.submit(segments, [(fi, ids[r].copy(), offs[r].copy()) for r, fi in meta.pop(k)]))

        try:
            self._run_batches((len(audio_paths) + Bs - 1) // Bs, fill, take, lang_id, threshold, pinned_in=pin, upload=upload)
            for f in futures:
                for fi, seg in f.result():
                    done[fi] = seg
        finally:
            post.shutdown(wait=True)
        return done

    def label_files(self, audio_paths, lang_id=None, confidence_threshold=0.0, verbose=True):
        """-> list (per file) of [(start_s, end_s, phoneme)] after merge + forced alignment."""
        if lang_id is not None and self.lang2id and lang_id > max(self.lang2id.values()):
            raise ValueError(f"Error: Language ID ({lang_id}) is higher than the latest ID ({max(self.lang2id.values())}) "
                             f"of this model.\n Languages and Codes available: {self.lang2id}")
        lang_name = self._lang_name(lang_id)
        decided_fast = {}                                 # file index -> segments of its one <= 30 s item, natively loaded
        if self.model.encoder_type == "whisper" and len(audio_paths) > 0:
            # only files whose header says 16 kHz are offered to the fast path (the others would be decoded there just to be turned
            # away, and their rows forwarded empty); a header that cannot be read leaves the decision to the loader
            cand = [fi for fi, p in enumerate(audio_paths) if A.wav_sample_rate(p) in (self.sr, None)]
            if cand:
                got = self._label_fast([audio_paths[fi] for fi in cand], lang_id, confidence_threshold, lang_name)
                decided_fast = {cand[j]: seg for j, seg in got.items()}
            # files at another rate: 16-bit PCM of at most 30 s goes to the GPU as it is and is resampled there (WFL_GPU_INGEST=0: host)
            if os.environ.get("WFL_GPU_INGEST", "1") != "0":
                by_rate = {}
                for fi, p in enumerate(audio_paths):
                    if fi in decided_fast:
                        continue
                    h = A.wav_header(p)
                    if h is None:
                        continue
                    tag, ch, sr, bits, nbytes = h
                    if tag == 1 and bits == 16 and ch in (1, 2) and sr != self.sr and sr > 0:
                        frames = nbytes // (2 * ch)
                        if 0 < frames and int(math.ceil(frames * self.sr / sr)) <= CHUNK_SAMPLES:
                            by_rate.setdefault(sr, []).append(fi)
                for sr, fis in by_rate.items():
                    got = self._label_resampled([audio_paths[fi] for fi in fis], sr, lang_id, confidence_threshold, lang_name)
                    decided_fast.update({fis[j]: seg for j, seg in got.items()})
        items, owner = [], []
        chunk_lens = []

        def load_one(path):
            chunks = A.load_items(path, self.sr)            # native: decode, resample, normalise, 30 s chunks (csrc/hostpost.hip)
            if chunks is None:                              # an encoding the native decoder does not take: the Python restatement
                audio = A.load_clip(path, self.sr)
                chunks = A.chunk_clip(audio, self.sr)
            return chunks

        slow = [fi for fi in range(len(audio_paths)) if fi not in decided_fast]
        # The native loader releases the GIL: files decode / resample on worker threads, all submitted at once; the items are
        # forwarded in waves, in file order, as their files arrive, so the forwards of one wave run under the loading of the next.
        decided = []
        pool = None
        try:
            if len(slow) > 1:
                from concurrent.futures import ThreadPoolExecutor
                pool = ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1, len(slow))))
                futs = [pool.submit(load_one, audio_paths[fi]) for fi in slow]
            wave = getattr(self, "_wave_items", None) or max(8 * self.batch_size, 64)   # items per wave: enough batches to fill
                                                                                        # the pipeline's slots (tests set a small one)
            start = 0
            for j, fi in enumerate(slow):
                chunks = futs[j].result() if pool is not None else load_one(audio_paths[fi])
                lens = [len(c) for c in chunks]
                if verbose and len(chunks) > 1:
                    print(f"Audio is too long ({sum(lens)/self.sr:.1f}s), splitting...")
                for c, n in zip(chunks, lens):
                    items.append(c)
                    owner.append(fi)
                    chunk_lens.append(n)
                if len(items) - start >= wave and j + 1 < len(slow):
                    decided.extend(self._forward_items(items[start:], lang_id, confidence_threshold))
                    start = len(items)
            if len(items) > start:
                decided.extend(self._forward_items(items[start:], lang_id, confidence_threshold))
        finally:
            if pool is not None:
                pool.shutdown(wait=True, cancel_futures=True)
        results = [[] for _ in audio_paths]
        clock = [0.0] * len(audio_paths)
        for (ids, offs), fi, n in zip(decided, owner, chunk_lens):
            s, e, ph = self._segments_of_item(ids, offs, lang_name)
            t0 = clock[fi]
            results[fi].append((s + t0, e + t0, ph))
            clock[fi] += n / self.sr
        names = self._names_for(lang_name)[1]
        final = []
        for fi, path in enumerate(audio_paths):
            if fi in decided_fast:                        # one <= 30 s item: decoded and merged by the post worker already
                segs = decided_fast[fi]
            else:
                parts = results[fi]
                s = np.concatenate([p[0] for p in parts]) if parts else np.empty(0)
                e = np.concatenate([p[1] for p in parts]) if parts else np.empty(0)
                ph = np.concatenate([p[2] for p in parts]).astype(np.int32) if parts else np.empty(0, np.int32)
                mode = self.config["postprocess"]["merge_segments"]
                if mode != "none" and s.size:
                    s, e, ph = npost.merge_segments(s, e, ph, mode)
                segs = npost.to_tuples(s, e, ph, names)
            forced = _read_forced(path, verbose)
            if forced is not None:
                aligned = pp.align_phoneme_list(segs, forced)
                if "SP" not in forced and "AP" not in forced and aligned:
                    before = [s for s in segs if s[2] in ("SP", "AP") and s[1] <= aligned[0][0]]
                    after = [s for s in segs if s[2] in ("SP", "AP") and s[0] >= aligned[-1][1]]
                    segs = before + aligned + after
                else:
                    segs = aligned
            final.append(segs)
        return final


def _read_forced(audio_path, verbose):
    """`{audio}.txt` forced phoneme list (infer.py:193, 210-215)."""
    txt = audio_path.replace(".wav", ".txt")
    if txt == audio_path or not os.path.exists(txt):
        return None
    forced = []
    with open(txt, "r", encoding="utf-8") as f:
        for line in f:
            forced.extend(line.strip().split())
    if verbose:
        print(f"Loaded forced phoneme list with {len(forced)} phonemes.")
    return forced


_LABELERS = {}


def _labeler(config_path, checkpoint_path, device):
    key = (os.path.abspath(str(config_path)), os.path.abspath(str(checkpoint_path)), str(device))
    if key not in _LABELERS:
        _LABELERS[key] = Labeler(config_path, checkpoint_path, device)
    return _LABELERS[key]


def _write_lab(path, segments):
    """save_lab (utils.py:76-81) with the text produced by the native formatter (wfl_host_format_lab: truncating int(t * 1e7))."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(npost.format_lab_tuples(segments))
    print(f"Predictions saved to: {path}")


def infer_audio(audio_path, config_path="config.yaml", checkpoint_path="best_model.pt", output_lab_path=None, device="cuda",
                lang_id=None, sample=False, top_k=0, top_p=0.0, temperature=1.0, confidence_threshold=0.0):
    lab = _labeler(config_path, checkpoint_path, device)
    segments = lab.label_files([audio_path], lang_id=lang_id, confidence_threshold=confidence_threshold)[0]
    if output_lab_path:
        if os.path.abspath(output_lab_path) == os.path.abspath(audio_path):
            # the reference would truncate the input WAV here (infer.py:410-411 + utils.py:77)
            output_lab_path = os.path.splitext(audio_path)[0] + ".lab"
        _write_lab(output_lab_path, segments)
    return segments


def infer_folder(folder_path: str, config_path: str = "config.yaml", checkpoint_path: str = "best_model.pt",
                 output_dir: str = "outputs", device: str = "cuda", lang_id: int = None, sample=False, top_k=0, top_p=0.0,
                 temperature=1.0, confidence_threshold=0.0):
    wav_files = sorted(f for f in os.listdir(folder_path) if f.lower().endswith(".wav"))
    os.makedirs(output_dir, exist_ok=True)
    # one process per GPU: every rank labels its own share of the files and writes its own .lab files (no collective)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        from .dist import shard_items
        sizes = [os.path.getsize(os.path.join(folder_path, f)) for f in wav_files]
        wav_files = [wav_files[i] for i in shard_items(sizes, world)[rank]]
    lab = _labeler(config_path, checkpoint_path, device)
    paths = [os.path.join(folder_path, f) for f in wav_files]
    all_segments = lab.label_files(paths, lang_id=lang_id, confidence_threshold=confidence_threshold) if paths else []
    for wav_file, segments in zip(wav_files, all_segments):
        print(f"\nInferencing: {wav_file}")
        _write_lab(os.path.join(output_dir, os.path.splitext(wav_file)[0] + ".lab"), segments)
        print("Predicted segments:")
        for start, end, ph in segments:
            print(f"({round(start, 2)}, {round(end, 2)}, {ph})")
    return all_segments


def main(argv=None):
    import click
    from pathlib import Path

    @click.command(help="Infer with WFL")
    @click.argument("path", metavar="PATH")
    @click.option("--checkpoint", "-ckpt", type=str, required=True, help="Path to WFL Checkpoint.")
    @click.option("--config", "-c", type=str, required=True, help="Path to Config file.")
    @click.option("--output", "-o", type=str, required=False, default=".", help="Path to output labels.")
    @click.option("--lang-id", "-l", type=int, required=False, default=None, help="Language ID.")
    @click.option("--sample", "-s", is_flag=True, help="Enable sampling instead of argmax")
    @click.option("--top-k", "-tk", type=int, default=0, help="Top-K sampling (range: 1-20)")
    @click.option("--top-p", "-tp", type=float, default=0.0, help="Top-P sampling (range: 0.1-1)")
    @click.option("--temperature", "-temp", type=float, default=1.0, help="Sampling temperature (range: 0.1-2)")
    @click.option("--device", "-d", type=str, default="auto", help='Device to use: "cuda", "cuda:0". Auto-detects if not specified.')
    @click.option("--confidence-threshold", "-ct", type=float, default=None,
                  help="Suppress predictions with low confidence. Set 0 to disable.")
    def cli(path, checkpoint, config, output, lang_id, sample, top_k, top_p, temperature, device, confidence_threshold):
        if sample:
            if top_k <= 0 and top_p <= 0.0:
                print("Sampling is enabled but neither --top-k nor --top-p is set.")
                sys.exit(1)
            if top_k > 0 and top_p > 0.0:
                print("You can't use both --top-k and --top-p at the same time.")
                sys.exit(1)
            if top_k < 0:
                print("top-k must be ≥ 1.")
                sys.exit(1)
            if top_p < 0.0 or top_p > 1.0:
                print("top-p must be between 0.1 and 1.0.")
                sys.exit(1)
            if temperature <= 0.0:
                print("temperature must be greater than 0.")
                sys.exit(1)
        requested = device.lower()
        if requested == "auto":
            device = "cuda"
        if not str(device).startswith("cuda") or not torch.cuda.is_available():
            print("This build runs on MI355X only (no CPU fallback): no ROCm device is available.", file=sys.stderr)
            sys.exit(1)
        inf_path = Path(path)
        cfg = load_config(Path(config))
        if confidence_threshold is None:
            confidence_threshold = cfg["postprocess"].get("confidence_threshold", 0.0)
        output_path = inf_path if output == "." else output
        if not inf_path.exists():
            print(f"Unable to locate folder {str(inf_path)}")
            sys.exit(1)
        if lang_id is not None and lang_id <= -1:
            lang_id = None
        kw = dict(config_path=str(config), checkpoint_path=str(checkpoint), device=device, lang_id=lang_id, sample=sample,
                  top_k=top_k, top_p=top_p, temperature=temperature, confidence_threshold=confidence_threshold)
        if inf_path.is_dir():
            infer_folder(folder_path=str(inf_path), output_dir=str(output_path), **kw)
        else:
            segments = infer_audio(audio_path=str(inf_path), output_lab_path=str(output_path), **kw)
            print("Predicted segments:")
            for start, end, ph in segments:
                print(f"({round(start, 2)}, {round(end, 2)}, {ph})")

    cli.main(args=argv, standalone_mode=True)


if __name__ == "__main__":
    main()
