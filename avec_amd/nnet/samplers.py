"""Length-bucketed batches for ragged utterances.

The reference pads every batch to its longest clip (nnet/collate_fn.py:143-146) and draws batches uniformly (functions.py:105-185): on an LRS2-shaped length
distribution 61 % of the frames a step computes are zero padding (padded-frame efficiency 0.39, tools/bench_variants.py lrs2_main).  This sampler keeps the
reference's epoch semantics -- every sample once per epoch, random order -- but forms batches from neighbours in LENGTH: the epoch's random permutation is cut
into windows of `window` samples, each window is sorted by length and cut into batches, and the batches are then visited in random order.  Randomness: which
samples share a window changes every epoch; inside a window batches are length-homogeneous.  Data parallel: every rank takes the batches rank, rank + world, ...
of the same (seeded) sequence, padded so that all ranks see the same number of batches."""
import torch


class LengthBucketBatchSampler(torch.utils.data.Sampler):
    def __init__(self, lengths, batch_size, window=None, shuffle=True, drop_last=True, seed=0, rank=0, world_size=1):
        self.lengths = [float(x) for x in lengths]
        self.batch_size, self.shuffle, self.drop_last, self.seed = int(batch_size), shuffle, drop_last, seed
        self.window = int(window) if window else 16 * self.batch_size
        self.rank, self.world_size, self.epoch = rank, world_size, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _batches(self):
        n = len(self.lengths)
        g = torch.Generator().manual_seed(self.seed + 7919 * self.epoch)
        order = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        batches = []
        for w in range(0, n, self.window):
            win = sorted(order[w:w + self.window], key=lambda i: self.lengths[i])
            for b in range(0, len(win), self.batch_size):
                batch = win[b:b + self.batch_size]
                if len(batch) == self.batch_size or not self.drop_last:
                    batches.append(batch)
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        if self.world_size > 1:
            rem = (-len(batches)) % self.world_size
            batches = batches + batches[:rem]               # (as DistributedSampler pads: every rank gets the same number of batches)
            batches = batches[self.rank::self.world_size]
        return batches

    def __iter__(self):
        return iter(self._batches())

    def __len__(self):
        return len(self._batches())

    def padded_frame_efficiency(self):
        """sum of the true lengths / sum over batches of (batch size x longest clip): what fraction of the computed frames is not padding"""
        num = den = 0.0
        for b in self._batches():
            num += sum(self.lengths[i] for i in b)
            den += len(b) * max(self.lengths[i] for i in b)
        return num / max(den, 1e-9)
