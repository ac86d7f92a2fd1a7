"""Input side of the training step: the reference's batch collation and a pinned-memory
host-to-device prefetcher.

`DataCollatorWithPadding` restates ref:train.py:90-133 (a class nested in `main()`, so it cannot be imported):
waveforms are right-padded with **-100** (the reference's choice - no attention mask ever reaches the speech encoder,
ref:speechmix/model.py:148, so the padded samples are convolved like audio; bug-compatible on purpose), labels are padded
by the tokenizer and the pad positions set to -100, a leading bos column shared by every row is cut.
Host-side only; nothing here touches the GPU kernels."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional, Union

import torch
from torch.nn.utils.rnn import pad_sequence


def _pad_ids(rows: List[List[int]], pad_id: int, multiple: Optional[int] = None, max_length: Optional[int] = None):
    """tokenizer.pad(..., padding=True) on plain id lists: right-pad to the longest row (or max_length), mask of real tokens."""
    n = max(len(r) for r in rows)                      # padding=True == 'longest': max_length is not used by HF either
    if multiple:
        n = (n + multiple - 1) // multiple * multiple
    ids = torch.full((len(rows), n), pad_id, dtype=torch.int64)
    mask = torch.zeros((len(rows), n), dtype=torch.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.as_tensor(r, dtype=torch.int64)
        mask[i, :len(r)] = 1
    return ids, mask


@dataclass
class DataCollatorWithPadding:
    """ref:train.py:90-133.  `tokenizer` needs `pad_token_id` and `bos_token_id` (an HF tokenizer, or any object with
    those two attributes - the padding itself is done here so that no tokenizer files are needed on the GPU box)."""
    tokenizer: Any
    padding: Union[bool, str] = True
    max_length: Optional[int] = None
    max_length_labels: Optional[int] = None
    pad_to_multiple_of: Optional[int] = None
    pad_to_multiple_of_labels: Optional[int] = None
    selftype: bool = False

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
        batch = {}
        batch["input_values"] = pad_sequence([torch.as_tensor(f["input_values"], dtype=torch.float32) for f in features],
                                             batch_first=True, padding_value=-100)
        pad_id = self.tokenizer.pad_token_id
        labels, mask = _pad_ids([list(f["labels"]) for f in features], pad_id, self.pad_to_multiple_of_labels)
        if "text_input_ids" in features[0]:
            batch["text_input_ids"], _ = _pad_ids([list(f["text_input_ids"]) for f in features], pad_id,
                                                  self.pad_to_multiple_of_labels)
        labels = labels.masked_fill(mask.ne(1), -100)
        bos = getattr(self.tokenizer, "bos_token_id", None)
        if bos and bool((labels[:, 0] == bos).all()):          # (a bos id of 0 never triggers the cut - as in the reference)
            labels = labels[:, 1:]
        batch["labels"] = labels
        return batch


def filter_by_length(lengths, max_input_length_in_sec, min_input_length_in_sec=1, sample_rate=16000):
    """Indices of the clips the reference keeps (ref:train.py:276-286): min * 16000 < samples < max * 16000, strict."""
    lo, hi = min_input_length_in_sec * sample_rate, max_input_length_in_sec * sample_rate
    return [i for i, n in enumerate(lengths) if lo < n < hi]


def length_grouped_indices(lengths, batch_size, mega_batch_mult=None, generator=None):
    """Order of the training set under the reference's `--group_by_length` (ref:train.py:176, 297 -> HF Trainer's
    LengthGroupedSampler, TF:trainer_pt_utils.py get_length_grouped_indices): a random permutation cut into mega-batches of
    `mega_batch_mult * batch_size` clips, each sorted by length (longest first), with the globally longest clip moved to
    the front (so an out-of-memory shows on step 1).  Consecutive `batch_size` slices are then batches of near-equal
    length: the -100 padding of the collator - which the speech encoder convolves like audio - stays minimal, and a rank's
    batch shapes repeat, so workspaces and kernel choices are reused."""
    n = len(lengths)
    if mega_batch_mult is None:
        mega_batch_mult = min(n // (batch_size * 4), 50) or 1
    perm = torch.randperm(n, generator=generator).tolist()
    mb = mega_batch_mult * batch_size
    megas = [sorted(perm[i:i + mb], key=lambda j: lengths[j], reverse=True) for i in range(0, n, mb)]
    if megas:
        k = max(range(len(megas)), key=lambda j: lengths[megas[j][0]])
        megas[0][0], megas[k][0] = megas[k][0], megas[0][0]
    return [j for m in megas for j in m]


def bucketed_batches(dataset, collator, batch_size, lengths=None, group_by_length=True, generator=None, drop_last=False,
                     rank=0, world=1, seed=0, epoch=0):
    """Collated batches of one epoch for this rank: length-grouped order (or a plain permutation), cut into global batches
    of `batch_size * world` clips of which rank r takes every world-th clip (`idx[rank::world]`, as HF's
    DistributedLengthGroupedSampler does: a global batch is sorted by length, so a contiguous split would hand rank 0 the
    longest clips of every step and make every step wait for it).  With world > 1 every rank must cut the SAME permutation:
    without an explicit generator one is derived from (seed, epoch), never from the rank's own global RNG."""
    n = len(dataset)
    if generator is None and world > 1:
        generator = torch.Generator()
        generator.manual_seed(int(seed) * 1000003 + int(epoch))
    if lengths is None:
        lengths = [len(dataset[i]["input_values"]) for i in range(n)]
    order = length_grouped_indices(lengths, batch_size * world, generator=generator) if group_by_length \
        else torch.randperm(n, generator=generator).tolist()
    gb = batch_size * world
    for i in range(0, n, gb):
        idx = order[i:i + gb]
        if len(idx) < gb and (drop_last or world > 1):
            break
        mine = idx[rank::world]
        yield collator([dataset[j] for j in mine])


class DevicePrefetcher:
    """Wraps an iterable of collated batches: each batch is staged in pinned host memory and copied to the device on a
    side stream while the previous step computes (the 20.5-MB waveform batch of config 2 is <1 % of a step over PCIe,
    DESIGN.md §6 - this hides it completely)."""

    @classmethod
    def from_dataset(cls, dataset, collator, batch_size, device, lengths=None, group_by_length=True, generator=None,
                     rank=0, world=1, seed=0, epoch=0):
        """Length-bucketed epoch (see `bucketed_batches`) behind the prefetcher."""
        return cls(bucketed_batches(dataset, collator, batch_size, lengths, group_by_length, generator, rank=rank,
                                    world=world, seed=seed, epoch=epoch), device)

    def __init__(self, batches: Iterable[Dict[str, torch.Tensor]], device):
        self.it, self.device = iter(batches), torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._next = None
        self._stage()

    def _stage(self):
        try:
            b = next(self.it)
        except StopIteration:
            self._next = None
            return
        if self.stream is None:
            self._next = b
            return
        with torch.cuda.stream(self.stream):
            self._next = {k: (v.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            for v in self._next.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
        out = self._next
        self._stage()
        return out
