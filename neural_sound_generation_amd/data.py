"""On-disk input format and batching of the reference, restated without its absent
dependencies (nnmnkwii, librosa, TF-1 hparams): CPU-side Python/numpy only, feeding the HIP path through pinned
host buffers.

Format (written by the reference's preprocessing, src/preprocess.py:32-35, src/ljspeech.py:91-102,
src/cmu_arctic.py:120-128): `<data_root>/train.txt`, one utterance per line, 4 or 5 fields separated by "|":

    <audio .npy> | <mel .npy> | <timesteps> | <text> [ | <speaker id> ]

audio: (timesteps,) array, mel: (frames, 80) float32 array, timesteps = frames * hop_size.

Pieces (names and behaviour follow src/dataloader.py:97-202,324-434):
  * `MelSpecDataSource` / `RawAudioDataSource` -- parse train.txt, lengths, speaker ids, the train / test split
    (sklearn's train_test_split with the reference's random_state=1234, test_size=0.05);
  * `PartialyRandomizedSimilarTimeLengthSampler` -- sort by length, shuffle inside groups of 32 batches, permute batches;
  * `collate_fn` -- random crop to `max_time_steps` aligned to the hop size, zero-pad to the longest item:
    returns the reference's 5-tuple (x, y, c, g, input_lengths) with c (B, 80, T) channel-first;
  * `get_data_loaders` -- {"train", "test"} loaders;  `DevicePrefetcher` -- pinned-memory, side-stream H2D copies so
    the next batch's mel is resident when the step starts (the timed path of bench.py assumes resident inputs).

parity unpinned: the reference's loader cannot be imported here (ordinary missing dependencies); tests/test_host_logic.py
checks the format invariants on a synthetic data root.
"""
from __future__ import annotations

import os
import random
from os.path import join
from typing import List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.sampler import Sampler

HOP_SIZE = 256          # src/main.py:167-170, src/hparams.py: frames -> samples
NUM_MELS = 80


class _NPYDataSource:
    """One column of train.txt as a list of .npy paths (src/dataloader.py:97-152)."""

    def __init__(self, data_root: str, col: int, speaker_id: Optional[int] = None, train: bool = True,
                 test_size: Optional[float] = 0.05, test_num_samples: Optional[int] = None, random_state: int = 1234):
        self.data_root, self.col = data_root, col
        self.speaker_id = speaker_id
        self.train, self.test_size, self.test_num_samples, self.random_state = train, test_size, test_num_samples, random_state
        self.lengths: List[int] = []
        self.multi_speaker = False
        self.speaker_ids: Optional[List[int]] = None
        self.paths = self.collect_files()

    def interest_indices(self, n: int) -> np.ndarray:
        from sklearn.model_selection import train_test_split
        indices = np.arange(n)
        test_size = self.test_num_samples / n if self.test_size is None else self.test_size
        train_idx, test_idx = train_test_split(indices, test_size=test_size, random_state=self.random_state)
        return train_idx if self.train else test_idx

    def collect_files(self) -> List[str]:
        with open(join(self.data_root, "train.txt"), "rb") as f:
            rows = [ln.decode("utf-8").rstrip("\n").split("|") for ln in f.readlines() if ln.strip()]
        if not rows:
            raise ValueError(f"{self.data_root}/train.txt is empty")
        if len(rows[0]) not in (4, 5):
            raise ValueError("train.txt: expected 4 or 5 fields separated by '|', got %d" % len(rows[0]))
        self.multi_speaker = len(rows[0]) == 5
        lengths = np.array([int(r[2]) for r in rows])
        paths = np.array([join(self.data_root, r[self.col]) for r in rows])
        speakers = np.array([int(r[-1]) for r in rows]) if self.multi_speaker else None
        if self.multi_speaker and self.speaker_id is not None:      # one speaker of a multi-speaker set
            keep = speakers == self.speaker_id
            paths, lengths, speakers = paths[keep], lengths[keep], None
            self.multi_speaker = False
        idx = self.interest_indices(len(paths))
        self.lengths = [int(v) for v in lengths[idx]]
        if speakers is not None:
            self.speaker_ids = [int(v) for v in speakers[idx]]
        return [str(p) for p in paths[idx]]

    def collect_features(self, path: str) -> np.ndarray:
        return np.load(path, allow_pickle=False)

    def __len__(self):
        return len(self.paths)


class RawAudioDataSource(_NPYDataSource):
    def __init__(self, data_root, **kwargs):
        super().__init__(data_root, 0, **kwargs)


class MelSpecDataSource(_NPYDataSource):
    def __init__(self, data_root, **kwargs):
        super().__init__(data_root, 1, **kwargs)


class PartialyRandomizedSimilarTimeLengthSampler(Sampler):
    """Sort by length, shuffle inside groups of `batch_group_size`, permute whole batches (src/dataloader.py:158-202):
    a batch holds clips of similar length, so little of it is padding."""

    def __init__(self, lengths, batch_size=16, batch_group_size=None, permutate=True):
        self.lengths, self.sorted_indices = torch.sort(torch.LongTensor(lengths))
        self.batch_size = batch_size
        if batch_group_size is None:
            batch_group_size = min(batch_size * 32, len(self.lengths))
            if batch_group_size % batch_size != 0:
                batch_group_size -= batch_group_size % batch_size
        self.batch_group_size = max(batch_group_size, 0)
        self.permutate = permutate

    def __iter__(self):
        indices = self.sorted_indices.clone().tolist()
        g = self.batch_group_size
        e = 0
        if g > 0:
            for i in range(len(indices) // g):
                s, e = i * g, (i + 1) * g
                chunk = indices[s:e]
                random.shuffle(chunk)
                indices[s:e] = chunk
            if self.permutate and e > 0:
                batches = [indices[i:i + self.batch_size] for i in range(0, e, self.batch_size)]
                random.shuffle(batches)
                indices[:e] = [i for b in batches for i in b]
        tail = indices[e:]
        random.shuffle(tail)
        indices[e:] = tail
        return iter(indices)

    def __len__(self):
        return len(self.sorted_indices)


class MelDataset(Dataset):
    """(raw audio or None, mel (frames, 80), speaker id or None) per utterance (src/dataloader.py:205-229)."""

    def __init__(self, mel: MelSpecDataSource, audio: Optional[RawAudioDataSource] = None):
        self.mel, self.audio = mel, audio
        self.multi_speaker = mel.multi_speaker
        self.lengths = mel.lengths

    def __getitem__(self, idx):
        c = self.mel.collect_features(self.mel.paths[idx])
        if c.ndim != 2 or c.shape[1] != NUM_MELS:
            raise ValueError(f"{self.mel.paths[idx]}: expected a (frames, {NUM_MELS}) array, got {c.shape}")
        x = self.audio.collect_features(self.audio.paths[idx]) if self.audio is not None else None
        g = self.mel.speaker_ids[idx] if self.multi_speaker else None
        return x, c.astype(np.float32, copy=False), g

    def __len__(self):
        return len(self.mel)


def ensure_divisible(length, divisible_by=HOP_SIZE, lower=True):
    if length % divisible_by == 0:
        return length
    return length - length % divisible_by if lower else length + (divisible_by - length % divisible_by)


class Collate:
    """collate_fn of src/dataloader.py:324-434 for the local-conditioning (mel) case with upsampled features: crop
    every utterance to at most `max_time_steps` samples = max_time_steps // hop frames at a random frame-aligned
    offset, zero-pad to the longest in the batch.  Returns (x, y, c, g, input_lengths):
      x (B, 1, Tx) float32 audio or None, y (B, Tx, 1) or None, c (B, 80, Tc) float32, g (B,) int64 or None,
      input_lengths (B,) int64 = audio samples per item (frames * hop when there is no audio)."""

    def __init__(self, max_time_steps: Optional[int] = None, hop_size: int = HOP_SIZE, frame_multiple: int = 1,
                 rng: Optional[np.random.RandomState] = None):
        self.max_time_steps, self.hop = max_time_steps, hop_size
        self.frame_multiple = frame_multiple          # e.g. 4: every clip's frame count is cut to a multiple of the model's stride
        self.rng = rng if rng is not None else np.random

    def __call__(self, batch):
        items = []
        for x, c, g in batch:
            frames = len(c)
            if x is not None and len(x) != frames * self.hop:
                raise ValueError("audio and mel lengths disagree: %d samples vs %d frames x hop %d" % (len(x), frames, self.hop))
            if self.max_time_steps is not None:
                max_frames = ensure_divisible(self.max_time_steps, self.hop, True) // self.hop
                if frames > max_frames:
                    s = self.rng.randint(0, frames - max_frames)      # (the reference's exclusive upper bound)
                    c = c[s:s + max_frames]
                    if x is not None:
                        x = x[s * self.hop:(s + max_frames) * self.hop]
            if self.frame_multiple > 1 and len(c) >= self.frame_multiple:
                keep = len(c) - len(c) % self.frame_multiple
                c = c[:keep]
                if x is not None:
                    x = x[:keep * self.hop]
            items.append((x, c, g))
        has_audio = items[0][0] is not None
        input_lengths = [len(x) if has_audio else len(c) * self.hop for x, c, _ in items]
        max_frames = max(len(c) for _, c, _ in items)
        c_batch = np.zeros((len(items), max_frames, NUM_MELS), dtype=np.float32)
        for i, (_, c, _) in enumerate(items):
            c_batch[i, :len(c)] = c
        c_t = torch.from_numpy(c_batch).transpose(1, 2).contiguous()             # (B, 80, T)
        x_t = y_t = None
        if has_audio:
            max_len = max(input_lengths)
            x_batch = np.zeros((len(items), max_len), dtype=np.float32)
            for i, (x, _, _) in enumerate(items):
                x_batch[i, :len(x)] = x
            x_t = torch.from_numpy(x_batch).unsqueeze(1).contiguous()             # (B, 1, T)
            y_t = torch.from_numpy(x_batch).unsqueeze(-1).contiguous()            # (B, T, 1)
        g_t = torch.LongTensor([g for _, _, g in items]) if items[0][2] is not None else None
        return x_t, y_t, c_t, g_t, torch.LongTensor(input_lengths)


def get_data_loaders(data_root: str, batch_size: int, max_time_steps: Optional[int] = None, speaker_id=None, num_workers: int = 2,
                     with_audio: bool = False, test_size=0.05, test_num_samples=None, random_state: int = 1234,
                     pin_memory: bool = True, frame_multiple: int = 1):
    """{"train": loader, "test": loader} over `data_root` (src/dataloader.py get_data_loaders).  The train loader uses
    the length-bucketed sampler; both crop / pad through `Collate`."""
    loaders = {}
    for phase in ("train", "test"):
        train = phase == "train"
        kw = dict(speaker_id=speaker_id, train=train, test_size=test_size, test_num_samples=test_num_samples, random_state=random_state)
        mel = MelSpecDataSource(data_root, **kw)
        audio = RawAudioDataSource(data_root, **kw) if with_audio else None
        ds = MelDataset(mel, audio)
        sampler = PartialyRandomizedSimilarTimeLengthSampler(ds.lengths, batch_size=batch_size) if train else None
        loaders[phase] = DataLoader(ds, batch_size=batch_size, sampler=sampler, shuffle=False, num_workers=num_workers,
                                    collate_fn=Collate(max_time_steps, frame_multiple=frame_multiple), pin_memory=pin_memory)
    return loaders


class DevicePrefetcher:
    """Wraps a loader of (x, y, c, g, input_lengths): copies the NEXT batch's c (and g) to the GPU on a side stream from
    pinned memory while the current step runs, and hands out batches whose tensors are already resident."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        x, y, c, g, lens = batch
        with torch.cuda.stream(self.stream):
            c = (c if c.is_pinned() else c.pin_memory()).to(self.device, non_blocking=True)
            if g is not None:
                g = g.to(self.device, non_blocking=True)
        return x, y, c, g, lens

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while True:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            cur = nxt
            cur[2].record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                yield cur
                return
            yield cur

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset


def write_synthetic_data_root(data_root: str, n_utts: int = 12, min_frames: int = 40, max_frames: int = 200, n_speakers: int = 0,
                              with_audio: bool = True, seed: int = 0) -> None:
    """A data root in the reference's format filled with random mels in [0, 1) (tests, smoke runs: there is no network
    for LJSpeech / CMU Arctic here)."""
    rs = np.random.RandomState(seed)
    os.makedirs(data_root, exist_ok=True)
    lines = []
    for i in range(n_utts):
        frames = int(rs.randint(min_frames, max_frames + 1))
        mel = rs.rand(frames, NUM_MELS).astype(np.float32)
        np.save(join(data_root, "synth-mel-%05d.npy" % i), mel, allow_pickle=False)
        if with_audio:
            np.save(join(data_root, "synth-audio-%05d.npy" % i), rs.uniform(-1, 1, frames * HOP_SIZE).astype(np.float32), allow_pickle=False)
        fields = ["synth-audio-%05d.npy" % i, "synth-mel-%05d.npy" % i, str(frames * HOP_SIZE), "utterance %d" % i]
        if n_speakers > 0:
            fields.append(str(int(rs.randint(0, n_speakers))))
        lines.append("|".join(fields))
    with open(join(data_root, "train.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


def synthetic_mel_batch(B: int, T: int, generator: torch.Generator, device, n_mels: int = 80) -> torch.Tensor:
    """(B, 1, n_mels, T) float32 in [0, 1] with the texture of a normalised mel spectrogram (the range of
    src/audio_tacotron.py:228-234): a few harmonic ridges per clip whose pitch and energy drift over the frames, a little noise.
    For convergence checks and demos without a corpus -- white noise (bench.py's timing input) cannot be
    reconstructed, so a loss on it says nothing.  No reference counterpart."""
    t = torch.linspace(0, 1, T, device=device).view(1, 1, T)
    h = torch.linspace(0, 1, n_mels, device=device).view(1, n_mels, 1)
    x = torch.zeros(B, n_mels, T, device=device)

    def u():
        return torch.rand(B, 1, 1, generator=generator, device=device)

    for _ in range(4):
        f0, drift, rate, phase = u() * 0.5 + 0.1, (u() - 0.5) * 0.3, u() * 6 + 1, u() * 6.28
        centre = f0 + drift * torch.sin(rate * 6.28 * t + phase)
        width = u() * 0.04 + 0.02
        energy = 0.5 + 0.5 * torch.sin(u() * 20 * t + phase)
        x = x + energy * torch.exp(-((h - centre) / width) ** 2)
    x = x + 0.05 * torch.randn(B, n_mels, T, generator=generator, device=device)
    return torch.sigmoid(3.0 * (x - 0.5)).unsqueeze(1).contiguous()

